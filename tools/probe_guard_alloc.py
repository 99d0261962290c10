"""Guarded-allocator run of one eager training iteration: does any kernel read outside the tensors it was handed?

torch is switched to tools/guard_alloc (one hipMalloc per tensor, 2 MiB of NaN bytes either side, zeroed body) BEFORE the
first device allocation; then one iteration (forward, loss, backward, fused AdamW; bf16 autocast) runs with a device-wide
sync and a scan of every live floating-point device tensor behind each C-ABI call. The first call after which a tensor
holds a non-finite value is named -- with a zeroed body and finite inputs the only source of NaN is a read from a guard.

    hipcc -O1 -shared -fPIC -o tools/guard_alloc/libguard_alloc.so tools/guard_alloc/guard_alloc.cpp
    python tools/probe_guard_alloc.py [tiny|medium|tsfb]
"""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, 'tools', 'guard_alloc', 'libguard_alloc.so')
alloc = torch.cuda.memory.CUDAPluggableAllocator(so, 'guard_malloc', 'guard_free')
torch.cuda.memory.change_current_allocator(alloc)

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['LAVILA_TEXT_STREAM'] = os.environ.get('LAVILA_TEXT_STREAM', '0')
from helpers import build_model                                     # noqa: E402
from lavila.models.loss import CLIPLoss                             # noqa: E402
from lavila_amd import _cabi as C                                   # noqa: E402
from oracle import oracle as O                                      # noqa: E402
from probe_nan_fence import CONFIGS, HOST_ONLY                      # noqa: E402  (same geometries)


def live_bad():
    bad = {}
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda and o.is_floating_point() and o.numel() and o.untyped_storage().nbytes() >= o.numel() * o.element_size():
                if not bool(torch.isfinite(o).all()):
                    bad[id(o)] = (tuple(o.shape), str(o.dtype), int((~torch.isfinite(o)).sum()))
        except Exception:
            pass
    return bad


class Watch:
    def __init__(self, real):
        self.real, self.n, self.seen, self.hits, self.on = real, 0, set(), [], False

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        if not name.startswith('lvl_') or any(h in name for h in HOST_ONLY):
            return fn

        def call(*a):
            rc = fn(*a)
            if self.on:
                torch.cuda.synchronize()
                bad = live_bad()
                new = {k: v for k, v in bad.items() if k not in self.seen}
                if new:
                    self.hits.append((self.n, name, list(new.values())[:4]))
                    self.seen |= set(new)
                self.n += 1
            return rc
        return call


def main(which):
    cfg = CONFIGS[which]
    torch.cuda.set_device(0)
    model = build_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(O.procedural_weights(shapes, seed=5))
    model.cuda().train()
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-3, fused=True)
    B = cfg['batch']
    video, tokens = O.synthetic_batch(B, cfg['frames'], cfg['img'], seed=41)
    tokens = tokens.clone()
    tokens[:, 1:30] = tokens[:, 1:30] % 510 + 1
    tokens[:, 0], tokens[:, 30] = 510, 511
    tokens[:, 31:] = 0
    video, tokens = video.cuda(), tokens.cuda()
    w = Watch(C.lib())
    C.lib = lambda: w
    for it in range(2):
        w.on = it == 1
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = crit(model(video, tokens, use_checkpoint=False, norm_embed=True))
        out['loss'].backward()
        torch.cuda.synchronize()
        bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        opt.step()
        torch.cuda.synchronize()
        bad_p = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        print(f'{which} iteration {it}: loss {float(out["loss"]):.5f}; non-finite gradients {len(bad_g)} {bad_g[:5]}; '
              f'non-finite parameters after the step {len(bad_p)} {bad_p[:5]}', flush=True)
    print(f'{which}: {w.n} kernel calls watched; calls after which a live tensor first held a non-finite value: {len(w.hits)}')
    for n, name, what in w.hits[:20]:
        print(f'    call {n:4d} {name}: {what}')
    return len(w.hits)


if __name__ == '__main__':
    sys.exit(1 if main(sys.argv[1] if len(sys.argv) > 1 else 'tiny') else 0)
