"""Times lvl_text_embed_fwd / _bwd against torch's nn.Embedding path at the benched text shape (256 captions, 32 of 77
positions, vocabulary 49408, width 512) for bench-like tokens (30 random ids per caption) and for ragged captions (lengths
5..32, the rest padding: thousands of duplicates of id 0)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lavila_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
B, L, ctx, V, W = 256, 32, 77, 49408, 512
g = torch.Generator().manual_seed(0)
table = (torch.randn(V, W, generator=g) * 0.02).to(dev).requires_grad_(True)
pos = (torch.randn(ctx, W, generator=g) * 0.01).to(dev).requires_grad_(True)
up = torch.randn(B, L, W, generator=g).to(dev).bfloat16()


def tokens(ragged):
    t = torch.zeros(B, ctx, dtype=torch.long)
    t[:, 0] = 49406
    if ragged:
        for b in range(B):
            n = int(torch.randint(5, 32, (1,), generator=g))
            t[b, 1:n] = torch.randint(1, 2000, (n - 1,), generator=g)      # a natural-language-like head of the vocabulary
            t[b, n] = 49407
    else:
        t[:, 1:31] = torch.randint(1, 49406, (B, 30), generator=g)
        t[:, 31] = 49407
    return t.to(dev)[:, :L]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for ragged in (False, True):
    text = tokens(ragged)

    def own():
        table.grad = pos.grad = None
        x = ops.text_embed(text, table, pos, torch.bfloat16)
        x.backward(up)

    def lib():
        table.grad = pos.grad = None
        x = (torch.nn.functional.embedding(text, table) + pos[:L]).to(torch.bfloat16)
        x.backward(up)

    print(f'{"ragged captions (padding id x%d)" % int((text == 0).sum()) if ragged else "bench tokens"}: forward + backward '
          f'own kernels {timed(own):.1f} us, torch (embedding + add + cast, sort-based backward) {timed(lib):.1f} us', flush=True)
