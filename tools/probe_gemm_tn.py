"""Probe: lvl_linear_tn (hand-written MFMA forward / input-gradient GEMM with fused epilogues) vs the library GEMM
(+ the separate element-wise kernels it replaces) at the bench shapes. Correctness first, then HIP-event timings."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402,F401  (replays the tuned hipBLASLt table for the library side, as bench.py does)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from lavila_amd import _cabi as C  # noqa: E402
from lavila_amd import ops  # noqa: E402

M_BIG = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 785
only = sys.argv[2:]


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def qgelu(u):
    return u * torch.sigmoid(1.702 * u)


def qgelu_grad(u):
    s = torch.sigmoid(1.702 * u)
    return s * (1 + 1.702 * u * (1 - s))


def check(M, N, K, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    w = (torch.randn(N, K, device='cuda', generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device='cuda', generator=g)
    ref = x.float() @ w.float().t()
    out = {}
    y = ops.linear_tn_raw(x, w, b, C.EPI_BIAS)
    out['bias'] = ((y.float() - (ref + b)).abs().max() / (ref + b).abs().max()).item()
    y0 = ops.linear_tn_raw(x, w, None, C.EPI_BIAS)
    out['nobias'] = ((y0.float() - ref).abs().max() / ref.abs().max()).item()
    a, u = ops.linear_tn_raw(x, w, b, C.EPI_BIAS_QUICKGELU)
    ur = (ref + b).bfloat16()
    out['gelu_u'] = ((u.float() - ur.float()).abs().max() / ur.float().abs().max()).item()
    out['gelu_a'] = ((a.float() - qgelu(u.float())).abs().max() / a.float().abs().max()).item()
    uin = torch.randn(M, N, device='cuda', generator=g).bfloat16()
    du, cs = ops.linear_tn_raw(x, w, None, C.EPI_QUICKGELU_BWD, aux_in=uin)
    dref = ref * qgelu_grad(uin.float())
    out['bwd_du'] = ((du.float() - dref).abs().max() / dref.abs().max()).item()
    out['bwd_colsum'] = ((cs - dref.sum(0)).abs().max() / dref.sum(0).abs().max()).item()
    bad = {k: v for k, v in out.items() if not v < 1.5e-2}
    print(f'check M={M} N={N} K={K}: ' + ' '.join(f'{k}={v:.1e}' for k, v in out.items()) + ('  <-- BAD' if bad else ''),
          flush=True)
    return not bad


ok = True
for (M, N, K) in [(256, 256, 64), (1000, 512, 128), (777, 256, 192), (4096, 768, 768), (2049, 2304, 768), (513, 768, 3072), (70001, 768, 768), (33000, 256, 128), (1800, 256, 64)]:
    ok &= check(M, N, K)
print('CORRECTNESS', 'OK' if ok else 'FAILED', flush=True)

SHAPES = {'qkv': (2304, 768), 'proj': (768, 768), 'fc1': (3072, 768), 'fc2': (768, 3072), 'dqkv': (768, 2304),
          'Lqkv': (3072, 1024), 'Lproj': (1024, 1024), 'Lfc1': (4096, 1024), 'Lfc2': (1024, 4096),
          'tqkv': (1536, 512), 'tproj': (512, 512), 'tfc1': (2048, 512), 'tfc2': (512, 2048)}
for name, (N, K) in SHAPES.items():
    if (only and name not in only) or (not only and name[0] in 'Lt'):
        continue
    M = M_BIG if name[0] != 't' else 256 * 32
    x = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
    b = torch.randn(N, device='cuda')
    bb = b.bfloat16()
    fl = 2.0 * M * N * K
    t_own = timeit(lambda: ops.linear_tn_raw(x, w, b, C.EPI_BIAS))
    t_lib = timeit(lambda: F.linear(x, w, bb))
    line = f'{name}: M={M} N={N} K={K}  own {t_own:.3f} ms ({fl / t_own / 1e9:.0f} TF/s)  library {t_lib:.3f} ms ({fl / t_lib / 1e9:.0f} TF/s)'
    if name in ('fc1', 'Lfc1', 'tfc1'):
        t_f = timeit(lambda: ops.linear_tn_raw(x, w, b, C.EPI_BIAS_QUICKGELU))
        t_l = timeit(lambda: ops.bias_quick_gelu(F.linear(x, w), b))
        line += f' | +gelu fused {t_f:.3f} ms  library+kernel {t_l:.3f} ms'
    if name in ('fc2', 'Lfc2', 'tfc2'):
        # backward of fc2 feeding the GELU backward: du[M,K] = (dy[M,N] . W2[N,K]) * g'(u)
        dy = torch.randn(M, N, device='cuda').bfloat16()
        wt = w.t().contiguous()           # [K, N]: the transposed copy
        u = torch.randn(M, K, device='cuda').bfloat16()

        def lib_bwd():
            da = F.linear(dy, wt)
            du = torch.empty_like(u)
            dbias = torch.empty(K, dtype=torch.float32, device='cuda')
            ws = C.workspace('bias_quickgelu_bwd', M, K, 'cuda')
            C.check(C.lib().lvl_bias_quickgelu_bwd(C.ptr(da), C.ptr(u), None, C.ptr(du), C.ptr(dbias), C.ptr(ws), M, K,
                                                   C.LVL_BF16, C.stream_ptr()), 'gelu_bwd')
            return du
        t_f = timeit(lambda: ops.linear_tn_raw(dy, wt, None, C.EPI_QUICKGELU_BWD, aux_in=u))
        t_l = timeit(lib_bwd)
        line += f' | dgrad+gelu_bwd fused {t_f:.3f} ms  library+kernel {t_l:.3f} ms'
    print(line, flush=True)
