"""Micro-probe: runs one C-ABI attention entry point in a loop on synthetic bf16 data (for rocprofv3 runs).
usage: python tools/probe_attn.py [space|time] [fwd|bwd] [B] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lavila_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'space'
what = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
F, N, H = int(os.environ.get('PROBE_F', '4')), int(os.environ.get('PROBE_N', '196')), int(os.environ.get('PROBE_H', '12'))
T, D = 1 + F * N, 64 * H
if 'PROBE_STREAM' in os.environ:      # 1: streaming space kernels for every shape, -1: never (round-3 resident kernels)
    from lavila_amd import _cabi as C
    C.lib().lvl_debug_space_stream(int(os.environ['PROBE_STREAM']))
if 'PROBE_STREAM_VARIANT' in os.environ:
    from lavila_amd import _cabi as C
    C.lib().lvl_debug_stream_variant(int(os.environ['PROBE_STREAM_VARIANT']))
g = torch.Generator(device='cuda').manual_seed(0)
qkv = (torch.randn(B, T, 3 * D, device='cuda', generator=g) * 1.0).bfloat16().requires_grad_(True)
dout = torch.randn(B, T, D, device='cuda', generator=g).bfloat16()
for it in range(iters + 3):
    if it == 3:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    if what == 'fwd':
        with torch.no_grad():
            o = ops.divided_attention(qkv, F, N, H, mode)
    else:
        o = ops.divided_attention(qkv, F, N, H, mode)
        o.backward(dout)
        qkv.grad = None
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
nbytes = B * (T * 3 * D + T * D) * 2 * (1 if what == 'fwd' else 3)   # fwd: qkv in + out; fwd+bwd: + (qkv,out,dout in; dqkv out)
print(f'{mode} {what} B={B}: {dt*1e3:.3f} ms/iter, {nbytes/dt/1e9:.0f} GB/s algorithmic')
