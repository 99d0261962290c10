"""Micro-probe: LayerNorm / bias+QuickGELU C-ABI kernels at the bench shapes (TSF-B, local batch B), bf16.
Prints ms per launch and algorithmic GB/s. usage: python tools/probe_rowops.py [B] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lavila_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows, D = B * 785, 768
dev = 'cuda'


def timeit(name, fn, nbytes):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f'{name:34s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.0f} GB/s')


x = torch.randn(rows, D, device=dev).bfloat16()
y = torch.randn(rows, D, device=dev).bfloat16()
dy = torch.randn(rows, D, device=dev).bfloat16()
dadd = torch.randn(rows, D, device=dev).bfloat16()
g = torch.ones(D, device=dev)
b = torch.zeros(D, device=dev)
yb = torch.zeros(D, device=dev)
E = rows * D * 2

h, _, mean, rstd = ops.layernorm_fwd_raw(x, None, None, g, b, 1e-5, False)
timeit('ln_fwd plain (1 in, 1 out)', lambda: ops.layernorm_fwd_raw(x, None, None, g, b, 1e-5, False), 2 * E)
timeit('ln_fwd add keep_sum (2 in, 2 out)', lambda: ops.layernorm_fwd_raw(x, y, yb, g, b, 1e-5, True), 4 * E)
timeit('ln_fwd add no sum (2 in, 1 out)', lambda: ops.layernorm_fwd_raw(x, y, yb, g, b, 1e-5, False), 3 * E)
timeit('ln_bwd plain (2 in, 1 out)', lambda: ops.layernorm_bwd_raw(dy, x, None, None, g, mean, rstd, None, False),
       3 * E)
timeit('ln_bwd dadd+dsum (3 in, 1 out)', lambda: ops.layernorm_bwd_raw(dy, x, None, None, g, mean, rstd, dadd, True),
       4 * E)
h2, _, mean2, rstd2 = ops.layernorm_fwd_raw(x, y, yb, g, b, 1e-5, False)
timeit('ln_bwd recompute (3 in, 1 out)', lambda: ops.layernorm_bwd_raw(dy, x, y, yb, g, mean2, rstd2, None, True),
       4 * E)

u = torch.randn(rows, 4 * D, device=dev).bfloat16().requires_grad_(True)
bias = torch.zeros(4 * D, device=dev, requires_grad=True)
da = torch.randn(rows, 4 * D, device=dev).bfloat16()
with torch.no_grad():
    timeit('bias_gelu fwd (1 in, 1 out)', lambda: ops.bias_quick_gelu(u, bias), 8 * E)
a = ops.bias_quick_gelu(u, bias)


def gelu_bwd():
    torch.autograd.grad(a, [u, bias], da, retain_graph=True)


timeit('bias_gelu bwd (2 in, 1 out)', gelu_bwd, 12 * E)
