"""Which device allocations made DURING GraphedTrainStep._capture (process-group path) land in the ordinary allocator pool
instead of the graphs' private pool? A tensor allocated there, baked into a captured kernel's arguments and freed afterwards is
a dangling pointer at replay (found by NaN-filling all cached free blocks between replays: the chain then turns NaN).
One process, world_size 1 over gloo; prints the Python stacks of the offending allocations."""
import os
import sys
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29655')
os.environ['LAVILA_TEXT_STREAM'] = '0'
from helpers import build_model                                     # noqa: E402
from lavila.models.loss import CLIPLoss                             # noqa: E402
from lavila_amd.graph_step import GraphedTrainStep                  # noqa: E402
from oracle import oracle as O                                      # noqa: E402

CFG = dict(img=32, patch=16, frames=2, dim=256, depth=2, heads=4, t_width=256, t_heads=4, t_layers=2, vocab=512,
           embed=64, batch=3, gated=False)
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group(os.environ.get('PROBE_BACKEND', 'gloo'), rank=0, world_size=1)
model = build_model(CFG)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(O.procedural_weights(shapes, seed=5))
model.cuda().train()
crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-3, fused=True, capturable=True)
B = CFG['batch']
step = GraphedTrainStep(model, crit, opt, (B, 3, CFG['frames'], CFG['img'], CFG['img']), (B, 77), dev)
video, tokens = O.synthetic_batch(B, CFG['frames'], CFG['img'], seed=40)
tokens = tokens.clone()
tokens[:, 1:31] = tokens[:, 1:31] % 510 + 1
tokens[:, 0], tokens[:, 31] = 510, 511
step(video, tokens)                                # eager first call
torch.cuda.synchronize()
torch.cuda.memory._record_memory_history(enabled='all', context='all', stacks='python', max_entries=200000)
step(video, tokens)                                # capture + first replay
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
segs = [(s['address'], s['address'] + s['total_size'], tuple(s.get('segment_pool_id', (0, 0)))) for s in snap['segments']]


def pool_of(addr):
    for lo, hi, pid in segs:
        if lo <= addr < hi:
            return pid
    return None


step_stream = step._stream.cuda_stream
sites = {}
for ev in snap['device_traces'][0]:
    if ev['action'] != 'alloc':
        continue
    pid = pool_of(ev['addr'])
    if pid != (0, 0):
        continue
    frames = [f for f in ev.get('frames', []) if '/lavila_amd/' in f['filename'] or '/tools/' in f['filename']]
    key = tuple(f"{os.path.basename(f['filename'])}:{f['line']} {f['name']}" for f in frames[:5])
    e = sites.setdefault(key, [0, 0, set()])
    e[0] += 1
    e[1] += ev['size']
    e[2].add(ev['stream'])
print('step stream', step_stream, 'comm stream', None if step._comm is None else step._comm.cuda_stream)
print('ordinary-pool allocations during the capture call (count, bytes, streams, innermost lavila_amd frames):')
for key, (n, nbytes, streams) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
    print(f'  {n:4d} x  {nbytes:10d} B  streams {sorted(streams)}  ' + ' <- '.join(key))
dist.destroy_process_group()
