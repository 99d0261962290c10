"""Same-box A/B of lvl_linear_tn's epilogues on the video tower's shapes (M = 200 960 rows): the base library
(tools/probes/ab/liblavila_hip_base.so, built from an earlier commit by tools/ab_library_swap.sh build <commit>) against
the tree's library, loaded side by side with ctypes, repetitions interleaved, HIP events, best of 5 x 3 launches.

    python tools/probe_gemm_epilogues.py

Epilogues 4 / 5 (QuickGELU + derivative, multiply-by-aux + column sums) exist in the new library only; the base runs 1 / 2
on the same shape. The variants of a repetition run in a random order (the first launch after a synchronisation finds the
chip cooler: a fixed order favours whoever goes first by 1-2 %); median and best of 9 repetitions x 3 launches.
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

BASE = os.path.join(ROOT, 'tools', 'probes', 'ab', 'liblavila_hip_base.so')
fns = {}
for name, p in (('base', BASE), ('new', C.LIB_PATH)):
    if os.path.exists(p):
        f = ctypes.CDLL(p).lvl_linear_tn
        f.restype, f.argtypes = C.SIGNATURES['lvl_linear_tn']
        fns[name] = f

M = 256 * 785
dev = torch.device('cuda', 0)
# name: (N, K, epilogue of the new library, epilogue of the base library)
SHAPES = {'qkv': (2304, 768, 0, 0), 'proj+res': (768, 768, 3, 3), 'fc1+gelu(u)': (3072, 768, 1, 1),
          'fc1+gelu(deriv)': (3072, 768, 4, 1), 'fc2+res': (768, 3072, 3, 3), "dx_fc2*gelu'(u)": (3072, 768, 2, 2),
          'dx_fc2*aux': (3072, 768, 5, 2), 'dx_qkv': (768, 2304, 0, 0), 'dx_fc1': (768, 3072, 0, 0), 'dx_proj': (768, 768, 0, 0)}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def bench():
    for name, (N, K, epi_new, epi_base) in SHAPES.items():
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
        b = torch.randn(N, device=dev) if epi_new not in (2, 5) else None
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        aux_out = torch.empty_like(y) if epi_new in (1, 4) else None
        aux_in = torch.randn(M, N, device=dev).bfloat16() if epi_new in (2, 3, 5) else None
        colsum = torch.empty(N, dtype=torch.float32, device=dev) if epi_new in (2, 5) else None
        nws = C.lib().lvl_workspace_floats(b'linear_tn', M, N)
        ws = torch.empty(max(int(nws), 1), dtype=torch.float32, device=dev) if epi_new in (2, 5) else None
        sched = torch.zeros(16, dtype=torch.int32, device=dev)

        def call(which, dyn=False):
            epi = epi_new if which == 'new' else epi_base
            rc = fns[which](P(x), P(w), P(b), P(y), P(aux_out), P(aux_in), P(colsum), P(ws), P(sched) if dyn else None,
                            M, N, K, epi, C.LVL_BF16, st)
            assert rc == 0, rc

        variants = [(k, False) for k in fns] + [('new', True)]
        times = {v: [] for v in variants}
        for rep in range(10):
            order = list(variants)
            random.shuffle(order)
            for v in order:
                which, dyn = v
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    call(which, dyn)
                e1.record()
                torch.cuda.synchronize()
                if rep > 0:
                    times[v].append(e0.elapsed_time(e1) / 3)
        fl = 2.0 * M * N * K
        parts = []
        for (which, dyn), t in times.items():
            tag = which + ('+dyn' if dyn else '')
            parts.append(f'{tag} median {statistics.median(t):.4f} best {min(t):.4f} ms ({fl / statistics.median(t) / 1e9:.0f} TF/s)')
        print(f'{name:18s} ' + '  '.join(parts), flush=True)


if __name__ == '__main__':
    random.seed(0)
    bench()
