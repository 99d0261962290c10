"""Times one time-attention backward INCLUDING the q third of the qkv-bias gradient (lvl_divided_attn_bwd_bias without a
column-sum token for the v third) at the TSF-B bench shape, for the three rider variants of the register-tiled time backward
kernels (lvl_debug_time_bwd_rider): 0 = no rider (a pass over the q third of dqkv afterwards), 1 = column sums in registers at
3 waves per SIMD (22 spilled registers), 2 = at 2 waves per SIMD (no spills). Also with the v third handed over as a token."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavila_amd import _cabi as C, ops      # noqa: E402

B, Fr, N, H = int(os.environ.get('PROBE_B', 256)), 4, 196, 12
D, T = 64 * H, 1 + Fr * N
dev = 'cuda'
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, T, 3 * D, generator=g) * 1.2).to(dev, torch.bfloat16)
dout = torch.randn(B, T, D, generator=g).to(dev, torch.bfloat16)
out, lse = ops.divided_attn_fwd_raw(qkv, Fr, N, H, C.ATTN_TIME)
tok = dout.float().sum((0, 1)).contiguous()
ref = None
for rider in (0, 1, 2, 0, 1, 2):
    C.check(C.lib().lvl_debug_time_bwd_rider(rider), 'rider')
    for with_tok in (False, True):
        dqkv = torch.empty_like(qkv)
        ws = C.workspace('divided_attn_bwd', B * H, T, dev)
        n2 = int(C.lib().lvl_divided_attn_bwd_bias_ws(B, Fr, N, H, C.ATTN_TIME, C.dtype_code(qkv)))
        ws2 = torch.empty(n2, dtype=torch.float32, device=dev)
        db = torch.empty(3 * D, dtype=torch.float32, device=dev)

        def call():
            C.check(C.lib().lvl_divided_attn_bwd_bias(C.ptr(qkv), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv), C.ptr(ws),
                                                      C.ptr(tok) if with_tok else None, C.ptr(db), C.ptr(ws2), B, Fr, N, H,
                                                      C.ATTN_TIME, C.dtype_code(qkv), C.stream_ptr()), 'bwd_bias')
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            call()
        e.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = db.clone()
        err = ((db - ref).abs().max() / ref.abs().max()).item()
        print(f'rider={rider} token={int(with_tok)}: {s.elapsed_time(e) / 20:.4f} ms per backward + bias gradient  (d bias vs first variant: {err:.1e})')
C.lib().lvl_debug_time_bwd_rider(-1)
