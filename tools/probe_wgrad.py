"""Probe: weight-gradient GEMM dW[N,K] = dY[M,N]^T X[M,K] with M = 200960 -- library default vs batched split-M."""
import sys, time, torch
M = 200960
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    dy = torch.randn(M, N, device='cuda', dtype=torch.bfloat16)
    x = torch.randn(M, K, device='cuda', dtype=torch.bfloat16)
    flops = 2.0 * M * N * K
    t = bench(lambda: dy.t() @ x)
    line = f'N={N} K={K}: default {t:.3f} ms ({flops/t/1e9:.0f} TF)'
    ref = (dy.t().float() @ x.float()) if N * K <= 768 * 768 else None
    for S in (8, 16, 32, 64):
        if M % S: continue
        dys, xs = dy.view(S, M // S, N), x.view(S, M // S, K)
        f = lambda: torch.bmm(dys.transpose(1, 2), xs).float().sum(0)
        t = bench(f)
        line += f' | S={S}: {t:.3f} ms ({flops/t/1e9:.0f} TF)'
    print(line, flush=True)
    if ref is not None:
        out = torch.bmm(dy.view(32, M // 32, N).transpose(1, 2), x.view(32, M // 32, K)).float().sum(0)
        print('   rel err of split vs f32 ref:', ((out - ref).norm() / ref.norm()).item(), ' default:', (((dy.t() @ x).float() - ref).norm() / ref.norm()).item())
