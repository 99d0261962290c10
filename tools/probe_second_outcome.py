"""Where does the TSF-B-geometry step's SECOND OUTCOME under a poisoned allocator come from?

Round 5 left this open: the poisoned-replay test at batch 4 x 224 px took, about one run in seven, losses 1.40463 / 1.57504
at calls 1 / 2 instead of 1.40535 / 1.58009 -- finite, the same digits every time. This probe repeats the test's five calls
(eager, capture + first replay, three replays) RUNS times in one process, with the poison placed in front of a chosen
subset of the calls and a chosen fill value, and snapshots every parameter after every call. A run that leaves the clean
one is attributed to the first call after which a parameter differs, and the differing tensors are listed (a parameter
that moved differently after call k had a different gradient in call k; a loss that differs with equal parameters is a
forward difference).

    python tools/probe_second_outcome.py [--runs 8] [--poison all|replays|eager|capture|none] [--fill nan|big|one|zero]
                                         [--batch 4] [--text-stream 0|1]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

ap = argparse.ArgumentParser()
ap.add_argument('--runs', type=int, default=8)
ap.add_argument('--poison', default='all')
ap.add_argument('--fill', default='nan')
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--text-stream', default=None)
ap.add_argument('--calls', type=int, default=3)
ap.add_argument('--eager-only', action='store_true', help='plain eager loop (no GraphedTrainStep)')
args = ap.parse_args()
if args.text_stream is not None:
    os.environ['LAVILA_TEXT_STREAM'] = args.text_stream

from helpers import build_model                                     # noqa: E402
from lavila.models.loss import CLIPLoss                             # noqa: E402
from lavila_amd.graph_step import GraphedTrainStep                  # noqa: E402
from oracle import oracle as O                                      # noqa: E402

CFG = dict(img=224, patch=16, frames=4, dim=768, depth=1, heads=12, t_width=512, t_heads=8, t_layers=1, vocab=512,
           embed=256, batch=args.batch, gated=False)
FILL = {'nan': float('nan'), 'big': 3.0e38, 'one': 1.0, 'zero': 0.0, 'neg': -3.0e38}[args.fill]
WHEN = {'all': lambda it: True, 'replays': lambda it: it >= 1, 'eager': lambda it: it == 0, 'capture': lambda it: it == 1,
        'replay2': lambda it: it == 2, 'none': lambda it: False}[args.poison]
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)


def poison(streams):
    torch.cuda.synchronize()
    for st in streams:
        with torch.cuda.stream(st):
            junk = [torch.full((n // 4,), FILL, device=dev) for n in
                    (256, 2048, 16384, 131072, 1 << 20, 1 << 22, 1 << 24, 1 << 26) for _ in range(12)]
            del junk
    torch.cuda.synchronize()


def run(poisoned):
    model = build_model(CFG)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(O.procedural_weights(shapes, seed=5))
    model.to(dev).train()
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    B = CFG['batch']
    if args.eager_only:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-3, fused=True)

        def step(video, tokens):
            opt.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out = crit(model(video.to(dev), tokens.to(dev), use_checkpoint=False, norm_embed=True))
            out['loss'].backward()
            opt.step()
            return out
        step._stream = torch.cuda.current_stream()
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-3, fused=True, capturable=True)
        step = GraphedTrainStep(model, crit, opt, (B, 3, CFG['frames'], CFG['img'], CFG['img']), (B, 77), dev)
    losses, snaps = [], []
    for it in range(args.calls):
        video, tokens = O.synthetic_batch(B, CFG['frames'], CFG['img'], seed=40 + it)
        tokens = tokens.clone()
        tokens[:, 1:31] = tokens[:, 1:31] % 510 + 1
        tokens[:, 0], tokens[:, 31] = 510, 511
        tokens[:, 32:] = 0
        if poisoned and WHEN(it):
            poison([torch.cuda.current_stream(), step._stream])
        losses.append(float(step(video, tokens)['loss']))
        torch.cuda.synchronize()
        snaps.append({n: p.detach().float().cpu().clone() for n, p in model.named_parameters()})
    return losses, snaps


def main():
    clean_l, clean_s = run(False)
    print(f'clean     {["%.5f" % v for v in clean_l]}', flush=True)
    again_l, again_s = run(False)
    same = all(torch.equal(a[k], b[k]) for a, b in zip(clean_s, again_s) for k in a)
    print(f'clean #2  {["%.5f" % v for v in again_l]}  parameters bit-equal to clean: {same}', flush=True)
    if not same:
        report(clean_s, again_s, clean_l, again_l)
    n_bad = 0
    for r in range(args.runs):
        l, s = run(True)
        d = max(abs(a - b) if a == a and b == b else float('inf') for a, b in zip(clean_l, l))
        tag = 'SAME' if d < 2e-5 else 'DIFFERENT'
        print(f'poisoned {r} ({args.poison}, {args.fill}) {["%.5f" % v for v in l]}  max |d loss| {d:.2e}  {tag}', flush=True)
        if d >= 2e-5 or not all(torch.equal(a[k], b[k]) for a, b in zip(clean_s, s) for k in a):
            n_bad += d >= 2e-5
            report(clean_s, s, clean_l, l)
    print(f'{n_bad} of {args.runs} poisoned runs left the clean losses', flush=True)


def report(ref_s, s, ref_l, l):
    for it, (a, b) in enumerate(zip(ref_s, s)):
        diff = []
        for k in a:
            if not torch.equal(a[k], b[k]):
                d = (a[k] - b[k]).abs()
                diff.append((float(d.max()), int((d > 0).sum()), a[k].numel(), k))
        print(f'    after call {it}: loss {l[it]:.6f} vs clean {ref_l[it]:.6f}; {len(diff)} of {len(a)} parameter tensors differ')
        for dm, cnt, n, k in sorted(diff, reverse=True)[:14]:
            print(f'        {k:48s} max |d| {dm:.3e}  {cnt}/{n} elements')
        if diff:
            first = sorted(diff, reverse=True)
            big = [x for x in first if x[0] > 1e-6]
            print(f'        ({len(big)} tensors with max |d| > 1e-6)')
            break


if __name__ == '__main__':
    main()
