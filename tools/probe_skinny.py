"""Times lvl_linear_skinny's tilings (lvl_debug_skinny_variant) on the decoder's Conv1D shapes, next to lvl_linear_tn and
the library GEMM: python tools/probe_skinny.py [--out file]. Variants (rows x columns per workgroup): 0 shipped; 1 32x32 paired k-steps; 2 64x32; 3 64x64; 4 64x32 paired; 5 16x32
paired; 6 16x16 paired; 7 32x64; 8 32x32; 9 16x16; 10 / 11 64x64 with 3 / 4 k-steps ahead; 12 the same paired; 13 64x32 with 4; 14-18 the LDS-staged
kernel: 128x128, 64x128, 64x64, 128x64 tiles; 18 = 64x64 with two K groups; 19-21 = 32x64, 32x64 with two K groups,
64x32 with two K groups."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavila_amd import _cabi as C  # noqa: E402
from lavila_amd import ops  # noqa: E402


def timed(fn, n=40):
    for _ in range(15):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us


def graph_timed(fn, n=40):
    """the same launch n times inside one captured hipGraph: no host launch gaps (what a decode step sees)"""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out')
    a = ap.parse_args()
    shapes = [(M, N, K) for M in (64, 640) for (N, K) in ((768, 768), (768, 3072), (3072, 768), (2304, 768), (50432, 768))]
    res = {}
    for M, N, K in shapes:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(M, K, generator=g).bfloat16().cuda()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        b = torch.randn(N, generator=g).cuda()
        y = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
        want = (x.float() @ w.float().t() + b)
        row = {}
        for v in range(22):
            C.lib().lvl_debug_skinny_variant(v)

            def run():
                C.check(C.lib().lvl_linear_skinny(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(y), M, N, K, -1, C.stream_ptr()), 'skinny')
            run()
            err = (y.float() - want).abs().max().item()
            row[f'skinny_v{v}'] = round(timed(run), 2)
            assert err < 0.1, (M, N, K, v, err)
        C.lib().lvl_debug_skinny_variant(0)
        row['linear_tn'] = round(timed(lambda: ops.linear_tn_raw(x, w, b)), 2)
        bb = b.bfloat16()
        row['library'] = round(timed(lambda: torch.nn.functional.linear(x, w, bb)), 2)

        def shipped():
            C.check(C.lib().lvl_linear_skinny(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(y), M, N, K, -1, C.stream_ptr()), 'skinny')
        row['in_graph_skinny_v0'] = round(graph_timed(shipped), 2)
        for v in (15, 16, 18, 19, 20, 21):
            C.lib().lvl_debug_skinny_variant(v)
            row[f'in_graph_skinny_v{v}'] = round(graph_timed(shipped), 2)
        C.lib().lvl_debug_skinny_variant(0)
        row['in_graph_library'] = round(graph_timed(lambda: torch.nn.functional.linear(x, w, bb)), 2)
        row['in_graph_linear_tn'] = round(graph_timed(lambda: ops.linear_tn_raw(x, w, b)), 2)
        row['weights_MB'] = round(N * K * 2 / 1e6, 2)
        res[f'M{M}_N{N}_K{K}'] = row
        print(f'M{M}_N{N}_K{K}', row, flush=True)
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
