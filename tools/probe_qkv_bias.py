import sys, time, torch
sys.path.insert(0, '.')
from lavila_amd import ops
M, D = 256 * 785, 768
dqkv = torch.randn(M, 3 * D, device='cuda').bfloat16(); dout = torch.randn(M, D, device='cuda').bfloat16()
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('own kernel      %.3f ms' % bench(lambda: ops._qkv_bias_grad(dqkv, dout, torch.float32)))
print('torch 2 sums    %.3f ms' % bench(lambda: (dqkv[:, :D].sum(0, dtype=torch.float32), dout.sum(0, dtype=torch.float32))))
print('torch full sum  %.3f ms' % bench(lambda: dqkv.sum(0, dtype=torch.float32)))
a = ops._qkv_bias_grad(dqkv, dout, torch.float32)
print('max err q', (a[:D] - dqkv[:, :D].float().sum(0)).abs().max().item(), 'v', (a[2*D:] - dout.float().sum(0)).abs().max().item(), 'k', a[D:2*D].abs().max().item())
