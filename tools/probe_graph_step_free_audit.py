"""Which device blocks are FREED between the start of GraphedTrainStep's capture call and the next replay, and where were they
allocated? (A tensor whose address a captured kernel holds must not be among them.) Single process, no process group."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['LAVILA_TEXT_STREAM'] = '0'
from helpers import build_model                                     # noqa: E402
from lavila.models.loss import CLIPLoss                             # noqa: E402
from lavila_amd.graph_step import GraphedTrainStep                  # noqa: E402
from oracle import oracle as O                                      # noqa: E402

CFG = dict(img=32, patch=16, frames=2, dim=256, depth=2, heads=4, t_width=256, t_heads=4, t_layers=2, vocab=512,
           embed=64, batch=3, gated=False)
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
torch.cuda.memory._record_memory_history(enabled='all', context='all', stacks='python', max_entries=400000)
model = build_model(CFG)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(O.procedural_weights(shapes, seed=5))
model.cuda().train()
crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-3, fused=True, capturable=True)
B = CFG['batch']
step = GraphedTrainStep(model, crit, opt, (B, 3, CFG['frames'], CFG['img'], CFG['img']), (B, 77), dev)


def batch(it):
    video, tokens = O.synthetic_batch(B, CFG['frames'], CFG['img'], seed=40 + it)
    tokens = tokens.clone()
    tokens[:, 1:31] = tokens[:, 1:31] % 510 + 1
    tokens[:, 0], tokens[:, 31] = 510, 511
    return video, tokens


float(step(*batch(0))['loss'])
torch.cuda.synchronize()
marker = torch.empty(1234567, device=dev)          # a recognisable allocation: the capture call starts behind it
del marker
float(step(*batch(1))['loss'])                     # capture + first replay
torch.cuda.synchronize()
marker2 = torch.empty(1234569, device=dev)
del marker2
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
tr = snap['device_traces'][0]
i0 = max(i for i, e in enumerate(tr) if e['action'] == 'alloc' and e['size'] in (1234567 * 4, ((1234567 * 4 + 511) // 512) * 512))
i1 = max(i for i, e in enumerate(tr) if e['action'] == 'alloc' and e['size'] in (1234569 * 4, ((1234569 * 4 + 511) // 512) * 512))
segs = [(s['address'], s['address'] + s['total_size'], tuple(s.get('segment_pool_id', (0, 0)))) for s in snap['segments']]


def pool_of(addr):
    for lo, hi, pid in segs:
        if lo <= addr < hi:
            return pid
    return ('gone',)


last_alloc = {}
for i, e in enumerate(tr[:i0]):
    if e['action'] == 'alloc':
        last_alloc[e['addr']] = e
    elif e['action'] in ('free_completed',):
        last_alloc.pop(e['addr'], None)
print('step stream', step._stream.cuda_stream)
rows = {}
for e in tr[i0:i1]:
    if e['action'] != 'free_completed':
        continue
    a = last_alloc.get(e['addr'])
    if a is None:
        continue                                   # allocated inside the window as well: a temporary of the capture call
    frames = [f for f in a.get('frames', []) if '/lavila_amd/' in f['filename'] or '/tools/' in f['filename'] or 'optim' in f['filename']]
    key = (a['stream'], pool_of(a['addr']), tuple(f"{os.path.basename(f['filename'])}:{f['line']} {f['name']}" for f in frames[:4]))
    r = rows.setdefault(key, [0, 0])
    r[0] += 1
    r[1] += a['size']
print('blocks allocated BEFORE the capture call and freed during it / before the next call:')
for (stream, pid, key), (n, nbytes) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f'  {n:4d} x {nbytes:10d} B  stream {stream}  pool {pid}  ' + ' <- '.join(key))
