import sys, torch
sys.path.insert(0, '.')
from lavila_amd import _cabi as C
DEV = 'cuda'
g = torch.Generator(device=DEV).manual_seed(1)
bad = 0
for (M, N, K, epi) in [(200960, 3072, 768, 4), (200960, 768, 768, 3), (200960, 768, 3072, 5), (70001, 2304, 768, 0)]:
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=DEV, generator=g) if epi != 5 else None
    aux_in = torch.randn(M, N, device=DEV, generator=g).bfloat16() if epi in (3, 5) else None
    sched = torch.zeros(16, dtype=torch.int32, device=DEV)
    def run(s):
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        aux_out = torch.empty_like(y) if epi == 4 else None
        colsum = torch.empty(N, dtype=torch.float32, device=DEV) if epi == 5 else None
        ws = C.workspace('linear_tn', M, N, DEV) if epi == 5 else None
        C.check(C.lib().lvl_linear_tn(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(y), C.ptr(aux_out), C.ptr(aux_in), C.ptr(colsum), C.ptr(ws), C.ptr(s), M, N, K, epi, C.LVL_BF16, C.stream_ptr()), 'tn')
        return [t for t in (y, aux_out, colsum) if t is not None]
    want = run(None)
    for mod in (0, 0, 7, 3):
        C.lib().lvl_debug_late_workgroups(mod)
        for rep in range(40):
            got = run(sched)
            if not all(torch.equal(a, c) for a, c in zip(got, want)):
                bad += 1
    C.lib().lvl_debug_late_workgroups(0)
    torch.cuda.synchronize()
    print((M, N, K, epi), 'mismatches so far', bad, 'sched zero', int(sched.abs().sum()) == 0, flush=True)
print('SOAK', 'OK' if bad == 0 else 'FAILED')
