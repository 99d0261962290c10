"""Timing-only comparison of variant builds of lvl_linear_tn (tools/probes/ab/liblavila_hip_<tag>.so, e.g. the GM_EXP builds
that drop the K loop's barriers / vmcnt waits -- their RESULTS ARE WRONG, only the time is of interest) against the tree's
library: plain epilogue, the video tower's shapes, random order inside a repetition, median of 9 x 3 launches.

    python tools/probe_gemm_variants.py exp1 exp2 exp3
"""
import ctypes
import os
import random
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lavila_amd import _cabi as C  # noqa: E402

fns = {}
for name, p in [('tree', C.LIB_PATH)] + [(t, os.path.join(ROOT, 'tools', 'probes', 'ab', f'liblavila_hip_{t}.so')) for t in sys.argv[1:]]:
    f = ctypes.CDLL(p).lvl_linear_tn
    f.restype, f.argtypes = C.SIGNATURES['lvl_linear_tn']
    fns[name] = f
M = 256 * 785
dev = torch.device('cuda', 0)
EPI = int(os.environ.get('PROBE_EPI', '0'))      # 0 plain; 4 QuickGELU + derivative (two result tensors)
SHAPES = {'qkv': (2304, 768), 'proj': (768, 768), 'fc1': (3072, 768), 'fc2': (768, 3072)} if EPI == 0 else {'fc1': (3072, 768)}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
random.seed(0)
for name, (N, K) in SHAPES.items():
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    aux = torch.empty_like(y) if EPI in (1, 4) else None
    if os.environ.get('PROBE_AUX_IS_Y') and aux is not None:
        aux = y          # timing experiment: the second tensor's stores hit the first tensor's lines (no extra bytes to HBM)
    times = {k: [] for k in fns}
    for rep in range(10):
        order = list(fns)
        random.shuffle(order)
        for k in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rc = fns[k](P(x), P(w), P(b), P(y), P(aux), None, None, None, None, M, N, K, EPI, C.LVL_BF16, st)
                assert rc == 0
            e1.record()
            torch.cuda.synchronize()
            if rep:
                times[k].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    print(f'{name:6s} ' + '  '.join(f'{k} {statistics.median(t):.4f} ms ({fl / statistics.median(t) / 1e9:.0f} TF/s)' for k, t in times.items()), flush=True)
