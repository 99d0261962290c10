"""Probe: Linear-layer GEMM layouts at the bench shapes (M = 256*785 rows, bf16) with TunableOp tuning ON.
forward x@W^T (TN), data gradient dy@W (NN) vs dy@Wt^T with a pre-transposed weight copy (TN), weight gradient
default vs split-row. Prints ms and TFLOP/s per variant.  usage: python tools/probe_gemm_layouts.py [M]"""
import os
import sys
import time

os.environ.setdefault('PYTORCH_TUNABLEOP_ENABLED', '1')
os.environ.setdefault('PYTORCH_TUNABLEOP_TUNING', '1')
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS', '15')
os.environ.setdefault('PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS', '5')
os.environ.setdefault('PYTORCH_TUNABLEOP_FILENAME', 'gpurun_out/tunableop_probe.csv')
os.environ.setdefault('PYTORCH_TUNABLEOP_VERBOSE', '0')
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 785


def bench(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, (N, K) in {'qkv': (2304, 768), 'proj': (768, 768), 'fc1': (3072, 768), 'fc2': (768, 3072)}.items():
    x = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(M, N, device='cuda', dtype=torch.bfloat16)
    W = torch.randn(N, K, device='cuda', dtype=torch.bfloat16) * 0.02
    Wt = W.t().contiguous()
    fl = 2.0 * M * N * K
    res = {}
    res['fwd  x@W^T (TN)'] = bench(lambda: F.linear(x, W))
    res['fwd  x@Wt  (NN)'] = bench(lambda: x @ Wt)
    res['dgrad dy@W  (NN)'] = bench(lambda: dy @ W)
    res['dgrad dy@Wt^T (TN)'] = bench(lambda: F.linear(dy, Wt))
    res['wgrad dy^T@x'] = bench(lambda: dy.t() @ x)
    for S in (16, 32):
        dys, xs = dy.view(S, M // S, N), x.view(S, M // S, K)
        res[f'wgrad split S={S}'] = bench(lambda: torch.bmm(dys.transpose(1, 2), xs).float().sum(0))
        res[f'wgrad^T split S={S}'] = bench(lambda: torch.bmm(xs.transpose(1, 2), dys).float().sum(0))
    print(f'--- {name}: N={N} K={K} ({fl / 1e12:.3f} TFLOP)', flush=True)
    for k, t in res.items():
        print(f'   {k:22s} {t:7.3f} ms  {fl / t / 1e9:6.0f} TF/s', flush=True)
