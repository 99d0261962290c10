"""Same-box A/B of the GEMM epilogue's store cache policy (gemm_tn_mfma.hip GM_STORE_POLICY: 0 plain, 1 nt, 2 sc1,
3 sc0 sc1): every variant library is loaded side by side with ctypes and times the video tower's GEMM shapes at
M = 200 960 rows (HIP events, interleaved repetitions).

    # build container: hipcc -DGM_STORE_POLICY=k -c gemm_tn_mfma.hip ...; link into tools/probes/ab/liblavila_hip_sp<k>.so
    python tools/probe_gemm_store_policy.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lavila_amd import _cabi as C  # noqa: E402

LIBS = {0: C.LIB_PATH}
for k in (1, 2, 3):
    p = os.path.join(ROOT, 'tools', 'probes', 'ab', f'liblavila_hip_sp{k}.so')
    if os.path.exists(p):
        LIBS[k] = p
NAMES = {0: 'plain', 1: 'nt', 2: 'sc1', 3: 'sc0 sc1'}
fns = {}
for k, p in LIBS.items():
    h = ctypes.CDLL(p)
    f = h.lvl_linear_tn
    f.restype, f.argtypes = C.SIGNATURES['lvl_linear_tn']
    fns[k] = f

M = 256 * 785
dev = torch.device('cuda', 0)
# name: (N, K, epilogue)
SHAPES = {'qkv': (2304, 768, 0), 'proj+res': (768, 768, 3), 'fc1+gelu': (3072, 768, 1), 'fc2+res': (768, 3072, 3),
          'dx_fc2*gelu\'': (3072, 768, 2), 'dx_qkv': (768, 2304, 0), 'dx_fc1': (768, 3072, 0)}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


for name, (N, K, epi) in SHAPES.items():
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev) if epi != 2 else None
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    aux_out = torch.empty_like(y) if epi == 1 else None
    aux_in = torch.randn(M, N, device=dev).bfloat16() if epi in (2, 3) else None
    colsum = torch.empty(N, dtype=torch.float32, device=dev) if epi == 2 else None
    nws = C.lib().lvl_workspace_floats(b'linear_tn', M, N)
    ws = torch.empty(max(int(nws), 1), dtype=torch.float32, device=dev) if epi == 2 else None

    def call(k):
        rc = fns[k](P(x), P(w), P(b), P(y), P(aux_out), P(aux_in), P(colsum), P(ws), None, M, N, K, epi, C.LVL_BF16, st)
        assert rc == 0, rc

    ref = None
    times = {k: [] for k in fns}
    for rep in range(6):
        for k in fns:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                call(k)
            e1.record()
            torch.cuda.synchronize()
            if rep > 0:
                times[k].append(e0.elapsed_time(e1) / 3)
            if rep == 0:
                if ref is None:
                    ref = y.clone()
                else:
                    if not torch.equal(ref, y):
                        print(f'  !! {name}: policy {k} changes the result (max |d| {(ref.float() - y.float()).abs().max().item():.3e})', flush=True)
    fl = 2.0 * M * N * K
    line = '  '.join(f'{NAMES[k]} {min(t):.4f} ms ({fl / min(t) / 1e9:.0f} TF/s)' for k, t in times.items())
    print(f'{name:14s} {line}', flush=True)
