"""Times the decoder's row kernels at the narrator's sizes (64 clips, 12 heads, 256 image tokens, width 768):
lvl_cross_attn_rows_fwd over qrep (captions per clip) x waves per workgroup, lvl_decode_self_attn, lvl_gated_add_layernorm.
Launches are timed back to back inside a captured hipGraph (50 per replay), i.e. without host launch gaps.
    python tools/probe_decode_kernels.py [--out file]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lavila_amd import _cabi as C  # noqa: E402


def graph_time(fn, n=50):
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
    for _ in range(4):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (4 * n) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out')
    a = ap.parse_args()
    ctx, H, Tk = 64, 12, 256
    D = H * 64
    res = {}
    # 12 layers' worth of distinct image keys / values (600 MB) so that every launch streams from HBM like a decode step
    kvs = [torch.randn(ctx, Tk, 2 * D, device='cuda').bfloat16() for _ in range(12)]
    for qrep in (1, 2, 4, 10, 20):
        q = torch.randn(ctx * qrep, D, device='cuda').bfloat16()
        out = torch.empty_like(q)
        for nw in ((0,) if qrep == 1 else (0, 4, 16)):       # 0 = shipped (MFMA kernel for qrep >= 2), n = VALU form with n waves
            C.lib().lvl_debug_cross_attn_waves(nw)
            it = [0]

            def run():
                kv = kvs[it[0] % 12]
                it[0] += 1
                C.check(C.lib().lvl_cross_attn_rows_fwd(C.ptr(q), C.ptr(kv), C.ptr(out), ctx * qrep, qrep, Tk, H, C.LVL_BF16,
                                                        C.stream_ptr()), 'cross')
            res[f'cross_attn_qrep{qrep}_waves{nw}'] = round(graph_time(run, 48), 2)
    C.lib().lvl_debug_cross_attn_waves(0)
    for rows in (64, 640):
        qkv = torch.randn(rows, 3 * D, device='cuda').bfloat16()
        cache = torch.randn(rows, 77, 2 * D, device='cuda').bfloat16()
        out = torch.empty(rows, D, device='cuda', dtype=torch.bfloat16)
        for p in (8, 40, 76):
            pos = torch.tensor([p], dtype=torch.int32, device='cuda')
            res[f'self_attn_rows{rows}_pos{p}'] = round(graph_time(lambda: C.check(C.lib().lvl_decode_self_attn(
                C.ptr(qkv), C.ptr(cache), C.ptr(pos), C.ptr(out), rows, 77, H, C.LVL_BF16, C.stream_ptr()), 'self')), 2)
        x = torch.randn(rows, D, device='cuda').bfloat16()
        y = torch.randn(rows, D, device='cuda').bfloat16()
        h = torch.empty_like(x)
        ga, be = torch.ones(D, device='cuda'), torch.zeros(D, device='cuda')
        res[f'gated_add_ln_rows{rows}'] = round(graph_time(lambda: C.check(C.lib().lvl_gated_add_layernorm(
            C.ptr(x), C.ptr(y), None, C.ptr(ga), C.ptr(be), 1e-5, C.ptr(x), C.ptr(h), rows, D, C.LVL_BF16, C.stream_ptr()), 'ln')), 2)
    print(json.dumps(res, indent=1))
    if a.out:
        with open(a.out, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
