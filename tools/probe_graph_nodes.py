"""Node census of the hipGraphs GraphedTrainStep captures (graph_step.graph_node_types): kernel / memcpy / memset counts per
geometry. A memset NODE is the thing to look for (unreliable under replay on this ROCm build: profiles/r05_graph_memset_nodes.txt).

    LAVILA_GRAPH_MEMSET_NODES=warn python tools/probe_graph_nodes.py [small|tsfb|long_text ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('LAVILA_GRAPH_MEMSET_NODES', 'warn')
from helpers import build_model                                     # noqa: E402
from lavila.models.loss import CLIPLoss                             # noqa: E402
from lavila_amd.graph_step import GraphedTrainStep                  # noqa: E402
from oracle import oracle as O                                      # noqa: E402

GEOM = {
    'small': (dict(img=32, patch=16, frames=2, dim=256, depth=2, heads=4, t_width=256, t_heads=4, t_layers=2, vocab=512,
                   embed=64, batch=3, gated=False), 31),
    'tsfb': (dict(img=224, patch=16, frames=4, dim=768, depth=1, heads=12, t_width=512, t_heads=8, t_layers=1, vocab=512,
                  embed=256, batch=4, gated=False), 31),
    'long_text': (dict(img=32, patch=16, frames=2, dim=256, depth=1, heads=4, t_width=256, t_heads=4, t_layers=1, vocab=512,
                       embed=256, batch=48, gated=False), 70),
}
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
for name in (sys.argv[1:] or list(GEOM)):
    cfg, eot = GEOM[name]
    model = build_model(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(O.procedural_weights(shapes, seed=5))
    model.to(dev).train()
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-3, fused=True, capturable=True)
    B = cfg['batch']
    step = GraphedTrainStep(model, crit, opt, (B, 3, cfg['frames'], cfg['img'], cfg['img']), (B, 77), dev)
    for it in range(3):
        video, tokens = O.synthetic_batch(B, cfg['frames'], cfg['img'], seed=40 + it)
        tokens = tokens.clone()
        tokens[:, 1:eot] = tokens[:, 1:eot] % 510 + 1
        tokens[:, 0], tokens[:, eot] = 510, 511
        tokens[:, eot + 1:] = 0
        loss = float(step(video, tokens)['loss'])
    torch.cuda.synchronize()
    print(f'{name}: loss {loss:.5f}; node types per caption bucket {step.node_types}', flush=True)
