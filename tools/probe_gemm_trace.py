"""Debug probe: builds gemm_tn_mfma.hip with -DGM_TRACE into a scratch .so and prints per-tile timelines
(wall-clock stamps of wave 0 / wave 4 of every persistent workgroup).
    python tools/probe_gemm_trace.py --build ; python tools/probe_gemm_trace.py <shape> [epilogue 0|1|3|4|5]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, 'tools', 'probes', 'libgemm_trace%s.so' % os.environ.get('GM_VARIANT', ''))   # probe builds stay out of the product lib dir
if '--build' in sys.argv:
    src = os.path.join(ROOT, 'lavila_amd', 'csrc')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                           '-DGM_TRACE', '-fno-slp-vectorize', *[a for a in sys.argv[1:] if a.startswith('-D')], os.path.join(src, 'gemm_tn_mfma.hip'), os.path.join(ROOT, 'tools', 'probes', 'trace_stub.hip'),
                           '-o', SO])
    print('built', SO)
    sys.exit(0)
import torch  # noqa: E402

lib = ctypes.CDLL(SO)
SHAPES = {'qkv': (2304, 768), 'proj': (768, 768), 'fc1': (3072, 768), 'fc2': (768, 3072), 'dqkv': (768, 2304)}
name = sys.argv[1] if len(sys.argv) > 1 else 'qkv'
EPI = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, K = SHAPES[name]
M = int(os.environ.get('PROBE_M', 256 * 785))
x = torch.randn(M, K, device='cuda').bfloat16()
w = (torch.randn(N, K, device='cuda') * K ** -0.5).bfloat16()
b = torch.randn(N, device='cuda')
y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
slab = 2 * ((M + 255) // 256) * N // 2 if EPI == 5 else 0          # int64 words of column partials in front of the stamps
trace = torch.zeros(slab + 256 * 2 * 128 + 8 * 2 * 256, dtype=torch.int64, device='cuda')
aux_out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16) if EPI in (1, 4) else None
aux_in = torch.randn(M, N, device='cuda').bfloat16() if EPI in (3, 5) else None
P = ctypes.c_void_p
PN = lambda t: P(t.data_ptr()) if t is not None else None
for _ in range(3):
    trace.zero_()
    rc = lib.lvl_linear_tn_trace_epi(P(x.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(y.data_ptr()), PN(aux_out), PN(aux_in),
                                     P(trace.data_ptr()), ctypes.c_int64(M), N, K, EPI, P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
fine = trace.cpu()[slab + 256 * 2 * 128:].view(8, 2, 256)
raw = trace.cpu()[slab:slab + 256 * 2 * 128].view(256, 2, 128)
cyc = raw[:, :, 64:].double()
t = raw[:, :, :64].double() * 10e-3      # 100 MHz ticks -> us
t0 = t[t > 0].min()
for wg in (0, 1, 8, 100, 255):
    for g in (0, 1):
        r = t[wg, g]
        n = int((r > 0).sum())
        r = r[:n] - t0
        ev = ['start', 'first'] + [e for i in range((n - 2) // 2) for e in (f'L{i}', f'E{i}')]
        print(f'wg {wg} grp {g}: ' + ' '.join(f'{e}={v:.1f}' for e, v in list(zip(ev, r.tolist()))[:14]), '... end', f'{r[-1]:.1f}')
loops, epis = [], []
for wg in range(256):
    r = t[wg, 0]
    n = int((r > 0).sum())
    for i in range((n - 2) // 2):
        prev = r[1] if i == 0 else r[2 * i + 1]
        loops.append((r[2 + 2 * i] - prev).item())
        epis.append((r[3 + 2 * i] - r[2 + 2 * i]).item())
loops, epis = torch.tensor(loops), torch.tensor(epis)
print(f'main loop per tile: mean {loops.mean():.2f} us (min {loops.min():.2f}, max {loops.max():.2f}); '
      f'epilogue: mean {epis.mean():.2f} us (min {epis.min():.2f} max {epis.max():.2f}); tiles {len(loops)}')
nn = int((t[0, 0] > 0).sum())
print('shader clock: %.3f GHz' % ((cyc[0, 0, nn - 1] - cyc[0, 0, 0]) / (t[0, 0, nn - 1] - t[0, 0, 0]) / 1e3).item())
print('kernel span', (t.max() - t0).item(), 'us; first-data latency mean', (t[:, 0, 1] - t[:, 0, 0]).mean().item())

if fine.abs().sum() > 0:
    # 4 stamps per phase: L-open (after the barrier that opens L), reads issued, C-open (after the C barrier), MFMAs issued
    for wg in (0, 3):
        for g in (0, 1):
            r = fine[wg, g]
            n = int((r > 0).sum())
            r = (r[:n] - r[0]).tolist()
            names = ['Lo', 'rd', 'Co', 'mf']
            print(f'wg {wg} grp {g} cycles:', ' '.join(f'{names[k % 4]}{r[k]}' for k in range(min(n, 36))))
    d = fine[:, :, :64].double()
    ph = d.view(8, 2, 16, 4)
    print('mean cycles  L-open->reads issued %.0f | reads issued->C-open (wait+barrier) %.0f | C-open->MFMAs issued %.0f | MFMAs issued->next L-open (wait+barrier) %.0f | phase %.0f'
          % ((ph[..., 1] - ph[..., 0]).mean(), (ph[..., 2] - ph[..., 1]).mean(), (ph[..., 3] - ph[..., 2]).mean(),
             (ph[:, :, 1:, 0] - ph[:, :, :-1, 3]).mean(), (ph[:, :, 1:, 0] - ph[:, :, :-1, 0]).mean()))
