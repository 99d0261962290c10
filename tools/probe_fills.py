"""Which host ops launch the small FillFunctor kernels of a training step? torch.profiler (CPU + GPU activities) around
one bench-like step; prints, per GPU kernel name containing 'Fill', the host ops that launched it.
usage: python tools/probe_fills.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

sys.argv = ['bench.py', '--no-cpu-baseline', '--no-events']
args = bench.parse()
device = torch.device('cuda', 0)
model = bench.build_model(args, device)
from lavila.models.loss import CLIPLoss  # noqa: E402
crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
decay = [p for n, p in model.named_parameters() if not (p.ndim < 2 or 'bias' in n or 'ln' in n or 'bn' in n)]
no_decay = [p for n, p in model.named_parameters() if (p.ndim < 2 or 'bias' in n or 'ln' in n or 'bn' in n)]
opt = torch.optim.AdamW([{'params': decay, 'weight_decay': 0.01}, {'params': no_decay, 'weight_decay': 0.0}],
                        lr=3e-5, betas=(0.9, 0.999), eps=1e-8, fused=True)
video, tokens = bench.synthetic(args, 0, device, 224)


def step():
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model(video, tokens.clone(), norm_embed=True)
        loss = crit(out)['loss']
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    model.logit_scale.data.clamp_(0, 4.6052)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
# map launch correlation -> innermost host op
agg = collections.Counter()
evs = prof.events()
for e in evs:
    if e.device_type.name in ('CUDA', 'PrivateUse1') or 'Fill' not in (e.name or ''):
        pass
for e in evs:
    for k in getattr(e, 'kernels', []) or []:
        if 'Fill' in k.name:
            st = [f for f in (e.stack or []) if ('lavila_amd' in f or 'bench' in f or 'optim' in f or 'probe_fills' in f)]
            agg[(k.name[:60], e.name, st[0][-90:] if st else '-')] += 1
for k, v in agg.most_common(30):
    print(v, k)
print('total fill kernels', sum(agg.values()))
