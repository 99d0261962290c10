"""Same-box timing of the space / time divided-attention entry points across variant builds of the library
(tools/probes/ab/liblavila_hip_<tag>.so, e.g. attention files compiled with another scheduling strategy) against the tree's:
lvl_divided_attn_fwd and lvl_divided_attn_bwd at the benched shape (B 256, 4 x 196 + 1 tokens, 12 heads, bf16), HIP events,
random order inside a repetition, median of 9 x 5 launches; results compared with the tree's bit for bit.

    python tools/probe_attn_variants.py tag1 tag2 ...
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

libs = {'tree': C.LIB_PATH}
for t in sys.argv[1:]:
    libs[t] = os.path.join(ROOT, 'tools', 'probes', 'ab', f'liblavila_hip_{t}.so')
H = {}
for k, p in libs.items():
    h = ctypes.CDLL(p)
    for name in ('lvl_divided_attn_fwd', 'lvl_divided_attn_bwd', 'lvl_workspace_floats'):
        f = getattr(h, name)
        f.restype, f.argtypes = C.SIGNATURES[name]
    H[k] = h
B, F, N, Hh = 256, 4, 196, 12
T, D = 1 + F * N, 64 * Hh
g = torch.Generator(device='cuda').manual_seed(0)
qkv = torch.randn(B, T, 3 * D, device='cuda', generator=g).bfloat16()
dout = torch.randn(B, T, D, device='cuda', generator=g).bfloat16()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
random.seed(0)
for mode, mname in ((0, 'space'), (1, 'time')):
    res, times = {}, {(k, w): [] for k in H for w in ('fwd', 'bwd')}
    for rep in range(10):
        order = list(times)
        random.shuffle(order)
        for k, w in order:
            h = H[k]
            out = torch.empty(B, T, D, dtype=torch.bfloat16, device='cuda')
            lse = torch.empty(B * Hh * T, dtype=torch.float32, device='cuda')
            wsf = torch.empty(max(int(h.lvl_workspace_floats(b'divided_attn_fwd', B * Hh, T)), 1), dtype=torch.float32, device='cuda')
            wsb = torch.empty(max(int(h.lvl_workspace_floats(b'divided_attn_bwd', B * Hh, T)), 1), dtype=torch.float32, device='cuda')
            dqkv = torch.empty_like(qkv)
            rc = h.lvl_divided_attn_fwd(P(qkv), P(out), P(lse), P(wsf), B, F, N, Hh, mode, C.LVL_BF16, st)
            assert rc == 0, rc
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                if w == 'fwd':
                    rc = h.lvl_divided_attn_fwd(P(qkv), P(out), P(lse), P(wsf), B, F, N, Hh, mode, C.LVL_BF16, st)
                else:
                    rc = h.lvl_divided_attn_bwd(P(qkv), P(out), P(dout), P(lse), P(dqkv), P(wsb), B, F, N, Hh, mode, C.LVL_BF16, st)
                assert rc == 0, rc
            e1.record()
            torch.cuda.synchronize()
            if rep:
                times[(k, w)].append(e0.elapsed_time(e1) / 5)
            res[(k, w)] = (out if w == 'fwd' else dqkv)
    for w in ('fwd', 'bwd'):
        print(f'{mname} {w}: ' + '  '.join(
            f'{k} {statistics.median(times[(k, w)]):.4f} ms' + ('' if k == 'tree' else (' (=)' if torch.equal(res[(k, w)], res[('tree', w)]) else ' (differs)'))
            for k in H), flush=True)
