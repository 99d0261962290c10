"""Bisect helper for whole-iteration hipGraph capture: python tools/probe_graph_step.py <stage> [textstream]
stage: fwd | fwdbwd | step | visual | text | loss ; prints 'OK <stage>' when capture + 2 replays succeed."""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
stage = sys.argv[1]
if len(sys.argv) > 2:
    os.environ['LAVILA_TEXT_STREAM'] = sys.argv[2]
import torch  # noqa: E402

from helpers import build_model  # noqa: E402
from lavila.models.loss import CLIPLoss  # noqa: E402
from lavila_amd import models as M  # noqa: E402

CFG = dict(img=224, patch=16, frames=4, dim=768, depth=2, heads=12, t_width=512, t_heads=8, t_layers=2, vocab=1024,
           embed=256, batch=4, gated=False)
torch.manual_seed(0)
model = build_model(CFG).cuda().train()
crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)
extra = stage.split('+')[1:]
stage = stage.split('+')[0]
if 'lr' in extra:
    for grp in opt.param_groups:
        grp['lr'] = torch.tensor(1e-4, device='cuda')
if stage == 'gstep':
    from lavila_amd.graph_step import GraphedTrainStep
    gs = GraphedTrainStep(model, crit, opt, tuple([CFG['batch'], 3, CFG['frames'], CFG['img'], CFG['img']]), (CFG['batch'], 77), 'cuda')
    v_h = torch.randn(CFG['batch'], 3, CFG['frames'], CFG['img'], CFG['img'])
    t_h = torch.randint(1, 1000, (CFG['batch'], 77)); t_h[:, 20] = 1023
    for i in range(4):
        if 'host' in extra:
            o = gs(v_h, t_h)
        else:
            o = gs(v_h.cuda(), t_h.cuda(), text_len=21)
        print('call', i, float(o['loss']), flush=True)
    print('OK gstep', extra, flush=True)
    sys.exit(0)
video = torch.randn(CFG['batch'], 3, CFG['frames'], CFG['img'], CFG['img'], device='cuda')
tokens = torch.randint(1, 1000, (CFG['batch'], 77), device='cuda')
tokens[:, 20] = 1023


def it():
    with M.fixed_text_length(24), torch.autocast('cuda', dtype=torch.bfloat16):
        if stage == 'visual':
            out = model.encode_image(video).float().sum()
        elif stage == 'text':
            out = model.encode_text(tokens).float().sum()
        else:
            o = model(video, tokens, use_checkpoint=False, norm_embed=True)
            out = crit(o)['loss'] if stage != 'fwd_noloss' else (o['image_embed'].sum() + o['text_embed'].sum())
    if stage in ('fwd', 'fwd_noloss'):
        return out
    out.backward()
    if stage == 'step':
        opt.step()
        if 'clamp' in extra:
            model.logit_scale.data.clamp_(0, 4.6052)
    if 'dict' in extra:
        return {'loss': out, 'x': out.detach() * 2}
    return out


opt.zero_grad(set_to_none=True)
it()
if stage == 'step':
    pass
opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print('eager ok', flush=True)
g = torch.cuda.CUDAGraph()
kw = {'pool': torch.cuda.graph_pool_handle()} if 'pool' in extra else {}
with torch.cuda.graph(g, **kw):
    out = it()
if isinstance(out, dict):
    out = out['loss']
print('captured', flush=True)
g.replay()
g.replay()
torch.cuda.synchronize()
print('OK', stage, sys.argv[2:], float(out), flush=True)
