"""Probe: lvl_linear_wgrad (hand-written MFMA weight gradient) vs the library paths at the bench shapes."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lavila_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 785
only = sys.argv[2:]          # optional layer names (qkv proj fc1 fc2)


def bench(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


SHAPES = {'qkv': (2304, 768), 'proj': (768, 768), 'fc1': (3072, 768), 'fc2': (768, 3072),
          'Lqkv': (3072, 1024), 'Lproj': (1024, 1024), 'Lfc1': (4096, 1024), 'Lfc2': (1024, 4096),
          'tqkv': (1536, 512), 'tproj': (512, 512), 'tfc1': (2048, 512), 'tfc2': (512, 2048)}
for name, (N, K) in SHAPES.items():
    if (only and name not in only) or (not only and name[0] in 'Lt'):
        continue
    x = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    dy = torch.randn(M, N, device='cuda', dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t_own = bench(lambda: ops.linear_wgrad_raw(dy, x, True))
    t_nob = bench(lambda: ops.linear_wgrad_raw(dy, x, False))
    S = 32 if N >= 3 * K else 16
    dys, xs = dy.view(S, M // S, N), x.view(S, M // S, K)
    t_lib = bench(lambda: (torch.bmm(dys.transpose(1, 2), xs).sum(0, dtype=torch.float32), dy.sum(0)))
    dw, db = ops.linear_wgrad_raw(dy, x, True)
    ref = torch.bmm(dys.transpose(1, 2), xs).sum(0, dtype=torch.float32)
    err = ((dw - ref).norm() / ref.norm()).item()
    print(f'{name}: N={N} K={K}  mfma {t_own:.3f} ms ({fl / t_own / 1e9:.0f} TF/s) no-dbias {t_nob:.3f} ms ({fl / t_nob / 1e9:.0f} TF/s)  library split+dbias {t_lib:.3f} ms '
          f'({fl / t_lib / 1e9:.0f} TF/s)   rel diff {err:.2e}', flush=True)
