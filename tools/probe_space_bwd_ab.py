"""Same-box A/B of the fused space-attention backward between two builds of the library (tools/ab_library_swap.sh build
<commit> -> tools/probes/ab/liblavila_hip_base.so vs lavila_amd/lib/liblavila_hip.so): lvl_divided_attn_bwd at the benched
shape (B 256, 4 frames, 196 locations, 12 heads, bf16), HIP events, alternating launches; results compared bit for bit.

    python tools/probe_space_bwd_ab.py [mode: space|time]
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lavila_amd import _cabi as C  # noqa: E402
from lavila_amd import ops  # noqa: E402

mode = 1 if (len(sys.argv) > 1 and sys.argv[1] == 'time') else 0
LIBS = {'base': os.path.join(ROOT, 'tools', 'probes', 'ab', 'liblavila_hip_base.so'), 'new': C.LIB_PATH}
fns = {}
for k, p in LIBS.items():
    h = ctypes.CDLL(p)
    f = h.lvl_divided_attn_bwd
    f.restype, f.argtypes = C.SIGNATURES['lvl_divided_attn_bwd']
    w = h.lvl_workspace_floats
    w.restype, w.argtypes = C.SIGNATURES['lvl_workspace_floats']
    fns[k] = (f, w)

B, F, N, H = int(os.environ.get('PROBE_B', '256')), 4, 196, 12
T, D = 1 + F * N, 64 * H
g = torch.Generator(device='cuda').manual_seed(0)
qkv = torch.randn(B, T, 3 * D, device='cuda', generator=g).bfloat16()
dout = torch.randn(B, T, D, device='cuda', generator=g).bfloat16()
out, lse = ops.divided_attn_fwd_raw(qkv, F, N, H, mode)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
res, times = {}, {k: [] for k in fns}
for rep in range(8):
    for k, (f, w) in fns.items():
        ws = torch.empty(int(w(b'divided_attn_bwd', B * H, T)), dtype=torch.float32, device='cuda')
        dqkv = torch.empty_like(qkv)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            rc = f(P(qkv), P(out), P(dout), P(lse), P(dqkv), P(ws), B, F, N, H, mode, C.LVL_BF16, st)
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        if rep:
            times[k].append(e0.elapsed_time(e1) / 5)
        res[k] = dqkv
print(f'{"time" if mode else "space"} backward incl. finalize, B={B}: ' +
      '  '.join(f'{k} {min(t):.4f} ms (median {sorted(t)[len(t) // 2]:.4f})' for k, t in times.items()), flush=True)
same = torch.equal(res['base'], res['new'])
d = (res['base'].float() - res['new'].float()).abs().max().item()
print(f'results bit-identical: {same} (max |d| {d:.3e})')
