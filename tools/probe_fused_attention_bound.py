"""Measured bound for the fused `norm1 -> qkv GEMM -> space attention -> proj` kernel SURVEY.md section 7 proposes
(q/k/v never touching HBM, one workgroup per (sample, frame) group of 197 tokens).

Such a kernel has to run the qkv and proj GEMMs on per-group row tiles: 197 valid rows in a 256-row MFMA tile (the
cls row belongs to every frame's key set but to no frame's query set, so a group is 196 query rows + the shared cls
row). This probe times the building blocks it would be made of, on the real shapes of BASELINE config 2:
  A  qkv + proj GEMMs on the dense token matrix, M = B*T = 200 960 rows              (what the step runs today)
  B  the same GEMMs on B*F = 1024 groups padded to 256 rows, M = 262 144 rows        (what a per-group kernel computes)
  C  the space attention forward kernel                                               (reads qkv once, writes o once)
and prints what fusing could save at most (the qkv write + read that disappear) against what the padding costs.
In TRAINING q/k/v are needed again by the backward, so the forward must still write them (or the backward must
recompute the qkv GEMM): only the forward's READ of qkv can disappear."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lavila_amd import _cabi as C  # noqa: E402
from lavila_amd import ops  # noqa: E402

B, Fr, N, H, D = 256, 4, 196, 12, 768
T = 1 + Fr * N


def timeit(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


wq = (torch.randn(3 * D, D, device='cuda') * D ** -0.5).bfloat16()
wp = (torch.randn(D, D, device='cuda') * D ** -0.5).bfloat16()
bq = torch.randn(3 * D, device='cuda')
bp = torch.randn(D, device='cuda')
res = {}
for tag, M in (('dense', B * T), ('padded', B * Fr * 256)):
    x = torch.randn(M, D, device='cuda').bfloat16()
    o = torch.randn(M, D, device='cuda').bfloat16()
    res[tag] = (timeit(lambda: ops.linear_tn_raw(x, wq, bq, C.EPI_BIAS)), timeit(lambda: ops.linear_tn_raw(o, wp, bp, C.EPI_BIAS)))
qkv = torch.randn(B, T, 3 * D, device='cuda').bfloat16()
t_attn = timeit(lambda: ops.divided_attn_fwd_raw(qkv, Fr, N, H, C.ATTN_SPACE))
qkv_bytes = B * T * 3 * D * 2
t_read = qkv_bytes / 4.6e12 * 1e3            # the streaming kernels of this repo sustain 4.5-4.8 TB/s
a, b = res['dense'], res['padded']
flops = 4 * Fr * N * (N + 1) * 64 * H * B     # useful QK^T + PV flops of the space groups
print(f'A dense   M={B * T}: qkv {a[0]:.3f} ms + proj {a[1]:.3f} ms = {sum(a):.3f} ms')
print(f'B padded  M={B * Fr * 256}: qkv {b[0]:.3f} ms + proj {b[1]:.3f} ms = {sum(b):.3f} ms   (+{sum(b) - sum(a):.3f} ms, x{sum(b) / sum(a):.2f})')
print(f'C space attention forward {t_attn:.3f} ms ({flops / t_attn / 1e9:.0f} TFLOP/s useful = {flops / t_attn / 1e9 / 2500:.1%} of the MFMA peak; '
      f'{(qkv_bytes * 4 / 3) / t_attn / 1e6:.0f} GB/s = {(qkv_bytes * 4 / 3) / t_attn / 1e6 / 8000:.1%} of HBM peak)')
print(f'fusing removes at most the forward read of qkv ({qkv_bytes / 1e9:.2f} GB = {t_read:.3f} ms at 4.6 TB/s) per layer; '
      f'the per-group row padding alone costs {sum(b) - sum(a):.3f} ms per layer')
print(f'MFMA fraction of the fused kernel on useful attention+qkv+proj flops, if everything else were free: '
      f'{(flops + 2.0 * B * T * D * 4 * D) / ((sum(b) + max(t_attn - t_read, 0)) * 1e9) / 2500:.1%}')
