"""Who issues the small fill / zero kernels of a training step? torch.profiler with stacks around one bench step.
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
opt = torch.optim.AdamW(model.parameters(), lr=3e-5, fused=True)
video, tokens = bench.synthetic(args, 0, device, 224)


def step():
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model(video, tokens.clone(), norm_embed=True)
        loss = crit(out)['loss']
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name in ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::zeros_like', 'aten::new_zeros', 'aten::full'):
        st = [f for f in (e.stack or []) if 'lavila_amd' in f or 'bench' in f or 'optim' in f or 'loss' in f]
        shape = tuple(e.input_shapes[0]) if e.input_shapes else ()
        agg[(e.name, st[0] if st else ((e.stack or ['?'])[0]), str(shape)[:40])] += 1
for k, v in agg.most_common(40):
    print(v, k)
