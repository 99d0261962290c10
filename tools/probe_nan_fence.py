"""NaN fence: does the result of any kernel depend on memory OUTSIDE the tensors it was handed?

One eager training iteration (forward, loss, backward; bf16 autocast as the pretraining loop runs it) is repeated once per
C-ABI call it makes; in repetition k every cached-but-free block of the allocator (and a fresh segment per size class) is
filled with NaN right in front of call k -- all the call's own tensors are live, so only memory the kernel must not read is
touched. A kernel that reads past its arguments (a tile tail that is masked on one operand only: 0 x NaN) turns the loss or
a gradient non-finite / different from the clean run, and is named. hipGraph replays make this matter: the memory next to a
graph's private pool belongs to whoever allocates next.

    python tools/probe_nan_fence.py [tiny|medium|tsfb]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['LAVILA_TEXT_STREAM'] = os.environ.get('LAVILA_TEXT_STREAM', '0')
from helpers import build_model                                     # noqa: E402
from lavila.models.loss import CLIPLoss                             # noqa: E402
from lavila_amd import _cabi as C                                   # noqa: E402
from oracle import oracle as O                                      # noqa: E402

CONFIGS = {
    'tiny': dict(img=32, patch=16, frames=2, dim=256, depth=2, heads=4, t_width=256, t_heads=4, t_layers=2, vocab=512,
                 embed=64, batch=3, gated=False),
    'medium': dict(img=80, patch=16, frames=4, dim=256, depth=1, heads=4, t_width=256, t_heads=4, t_layers=1, vocab=512,
                   embed=64, batch=5, gated=False),
    'tsfb': dict(img=224, patch=16, frames=4, dim=768, depth=1, heads=12, t_width=512, t_heads=8, t_layers=1, vocab=512,
                 embed=256, batch=3, gated=False),
    # the geometry of tests/test_gpu_graph_step.py's poisoned-replay test (batch 4: 16 space groups, 3140 token rows)
    'tsfb4': dict(img=224, patch=16, frames=4, dim=768, depth=1, heads=12, t_width=512, t_heads=8, t_layers=1, vocab=512,
                  embed=256, batch=4, gated=False),
    # two video blocks: the full chain (residual epilogues, pending MLP, tokens) in front of the cls-only last block
    'tsfb4x2': dict(img=224, patch=16, frames=4, dim=768, depth=2, heads=12, t_width=512, t_heads=8, t_layers=2, vocab=512,
                    embed=256, batch=4, gated=False),
}
HOST_ONLY = ('_ws', 'workspace', 'last_error', 'lvl_set_', 'lvl_debug', '_rows', '_floats', 'lvl_version')

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)


def poison():
    """NaN into every free cached block of the current stream's pools + one fresh segment per size class."""
    torch.cuda.synchronize()
    junk = []
    for nbytes in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10, 512):
        for _ in range(4000):
            before = torch.cuda.memory_reserved()
            junk.append(torch.empty(nbytes // 4, dtype=torch.float32, device=dev))
            if torch.cuda.memory_reserved() > before:
                break                                   # that one came from the driver: the free lists of this class are empty
    for j in junk:
        j.fill_(float('nan'))
    del junk
    torch.cuda.synchronize()


class Fenced:
    def __init__(self, real):
        self.real, self.calls, self.at = real, [], None

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        if not name.startswith('lvl_') or any(h in name for h in HOST_ONLY):
            return fn

        def call(*a):
            if self.at == len(self.calls):
                poison()
            self.calls.append(name)
            return fn(*a)
        return call


def main(which):
    cfg = CONFIGS[which]
    model = build_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(O.procedural_weights(shapes, seed=5))
    model.cuda().train()
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    B = cfg['batch']
    video, tokens = O.synthetic_batch(B, cfg['frames'], cfg['img'], seed=41)
    tokens = tokens.clone()
    tokens[:, 1:30] = tokens[:, 1:30] % 510 + 1
    tokens[:, 0], tokens[:, 30] = 510, 511
    tokens[:, 31:] = 0
    video, tokens = video.cuda(), tokens.cuda()
    proxy = Fenced(C.lib())
    C.lib = lambda: proxy

    graph_paths = os.environ.get('FENCE_GRAPH_PATHS') == '1'
    if graph_paths:
        # the branches the code takes under hipGraph capture (weight copies re-cast per use, tile counters allocated per
        # launch, caption length fixed by the caller), in an eager iteration where the fence can look at every call
        torch.cuda.is_current_stream_capturing = lambda *a, **k: True
    from lavila_amd import models as M
    import contextlib

    def iteration(at):
        proxy.calls, proxy.at = [], at
        for p in model.parameters():
            p.grad = None
        with (M.fixed_text_length(32) if graph_paths else contextlib.nullcontext()):
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out = crit(model(video, tokens, use_checkpoint=False, norm_embed=True))
            out['loss'].backward()
        torch.cuda.synchronize()
        return out['loss'].detach().float().cpu(), {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()
                                                     if p.grad is not None}

    iteration(None)
    loss0, g0 = iteration(None)
    loss1, g1 = iteration(None)
    noise = max(((g0[k] - g1[k]).abs().max() / (g0[k].abs().max() + 1e-30)).item() for k in g0)
    names = list(proxy.calls)
    print(f'{which}: {len(names)} kernel calls per iteration; run-to-run noise of the clean iteration {noise:.1e}', flush=True)
    bad = []
    for k, name in enumerate(names):
        loss, g = iteration(k)
        assert proxy.calls == names
        worst, where = 0.0, ''
        if not torch.isfinite(loss):
            worst, where = float('inf'), 'loss'
        for n in g0:
            d = (g[n] - g0[n]).abs().max().item() / (g0[n].abs().max().item() + 1e-30)
            if not (d <= worst):          # NaN compares false: caught here
                worst, where = (float('inf') if d != d else d), n
        if not (worst <= max(10 * noise, 1e-6)):
            bad.append((k, name, worst, where))
            print(f'  call {k:4d} {name:34s} reads outside its arguments: worst change {worst:.2e} ({where})', flush=True)
    print(f'{which}: {len(bad)} of {len(names)} calls depend on memory outside their arguments', flush=True)
    by = {}
    for k, name, w, where in bad:
        by.setdefault(name, []).append(k)
    for name, ks in by.items():
        print(f'    {name}: calls {ks[:12]}{" ..." if len(ks) > 12 else ""}')
    return len(bad)


if __name__ == '__main__':
    sys.exit(1 if main(sys.argv[1] if len(sys.argv) > 1 else 'tiny') else 0)
