"""Per-rank cost of the contrastive head at BASELINE configs[2] (W = 8 ranks x B = 256 -> G = 2048, E = 256), on ONE GPU:
the slab formulation (this rank's two [B, G] slabs: lvl_clip_loss_fwd + lvl_clip_loss_bwd) next to what every rank
of the reference's vissl path computes (the full G x G problem, loss.py:74-110) -- the O(G^2 / W) vs O(G^2) claim of
DESIGN.md section 5, measured. usage: python tools/probe_clip_loss.py [B] [W] [E]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lavila_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
E = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = 'cuda'


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n          # us


print(f'contrastive head, local batch B={B}, E={E}; slab = this rank\'s rows of logits_per_image and logits_per_text')
print(f'{"W":>3} {"G":>6} {"dtype":>6} | {"slab fwd us":>11} {"slab bwd us":>11} {"fwd+bwd":>9} | {"full GxG fwd":>12} {"full bwd":>9} '
      f'{"full f+b":>9} | slab/full')
for dt in (torch.float32, torch.bfloat16):
    for w in sorted({1, 2, 4, W}):
        G = w * B
        g = torch.Generator(device=dev).manual_seed(G)
        img = torch.nn.functional.normalize(torch.randn(G, E, device=dev, generator=g), dim=-1).to(dt)
        txt = torch.nn.functional.normalize(torch.randn(G, E, device=dev, generator=g), dim=-1).to(dt)
        scale = torch.tensor([14.2857], device=dev)
        up = torch.ones(1, device=dev)
        row0 = (w - 1) * B

        def fwd(b, r0):
            return ops.clip_loss_fwd_raw(img, txt, scale, b, r0)
        stats, _, _ = fwd(G, 0)
        lse_all = stats[..., 0].contiguous()

        def bwd(b, r0):
            return ops.clip_loss_bwd_raw(img, txt, lse_all, scale, up, 1.0 / (2 * G), b, r0)
        tf, tb = timeit(lambda: fwd(B, row0)), timeit(lambda: bwd(B, row0))
        ff, fb = timeit(lambda: fwd(G, 0)), timeit(lambda: bwd(G, 0))
        print(f'{w:>3} {G:>6} {str(dt).split(".")[1]:>6} | {tf:>11.1f} {tb:>11.1f} {tf + tb:>9.1f} | {ff:>12.1f} {fb:>9.1f} '
              f'{ff + fb:>9.1f} | {(tf + tb) / (ff + fb):.3f}', flush=True)
print('(the step of config 2/3 takes ~180 ms per rank: the head is < 0.2 % of it at every W; the slab cost grows '
      'with G, the replicated cost with G^2)')
