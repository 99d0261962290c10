"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.pt by running the UNMODIFIED reference
(/root/reference, imported through oracle/ref_import.py) on CPU in fp32.

Run in the build container (the reference is not present on the GPU box):
    python -m oracle.gen_golden            # writes tests/golden/*.pt

Weights and inputs are *procedural* (oracle.procedural_weights / oracle.synthetic_batch): fixtures
store only shapes, seeds and the reference's outputs, so they stay small enough to commit while
still pinning full-size configurations (config 1 of BASELINE.json: TSF-B/16, 2 frames of 112^2,
batch 4).
"""
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O          # noqa: E402
from oracle.ref_import import load_reference  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

CONFIGS = {
    # name: dict(model kwargs..., batch)
    'tiny_p16': dict(img=32, patch=16, frames=2, dim=128, depth=2, heads=2, t_width=128, t_heads=2,
                     t_layers=2, vocab=512, embed=64, batch=3, gated=False),
    'tiny_p14_gated': dict(img=42, patch=14, frames=3, dim=128, depth=2, heads=2, t_width=64, t_heads=1,
                           t_layers=1, vocab=512, embed=32, batch=2, gated=True),
    # 16-frame clips (the frame count of BASELINE.json configs[2]): F=16 time attention, T = 1 + 16*4
    'tiny_f16': dict(img=32, patch=16, frames=16, dim=128, depth=2, heads=2, t_width=128, t_heads=2,
                     t_layers=2, vocab=512, embed=64, batch=2, gated=False),
    # BASELINE.json configs[0]: CLIP_OPENAI_TIMESFORMER_BASE shape, 2 frames 112^2, batch 4
    'config1_tsfb_112': dict(img=112, patch=16, frames=2, dim=768, depth=12, heads=12, t_width=512,
                             t_heads=8, t_layers=12, vocab=49408, embed=256, batch=4, gated=False),
    # (round 4's format-1 fixtures of these shapes -- config2_tsfb_224_b8, tsfl14_224_b2, tsfl14_336_b2 -- were replaced by
    # the *_spread fixtures below in round 5 and are no longer generated)
    # round 5: "spread" fixtures (oracle.synthetic_batch / procedural_weights with spread=True: samples that do NOT
    # collapse onto one embedding, ragged captions, attention scores of a few units) with every 1-D gradient of six
    # blocks + row slices of their weight gradients stored in full (fixture format 2, compared per tensor on the
    # tensor's own scale). BASELINE configs[1]'s clip shape, both TSF-L/14 shapes, and the true 16-frame shapes of
    # configs[2] (TSF-B/16, 16 x 224^2: T = 3137) and configs[3] (TSF-L/14 at 336, 16 frames: T = 9217).
    'config2_tsfb_224_b8_spread': dict(img=224, patch=16, frames=4, dim=768, depth=12, heads=12, t_width=512,
                                       t_heads=8, t_layers=12, vocab=49408, embed=256, batch=8, gated=False, spread=True),
    'tsfl14_224_b2_spread': dict(img=224, patch=14, frames=4, dim=1024, depth=24, heads=16, t_width=768,
                                 t_heads=12, t_layers=12, vocab=49408, embed=256, batch=2, gated=False, spread=True),
    'tsfl14_336_b2_spread': dict(img=336, patch=14, frames=2, dim=1024, depth=24, heads=16, t_width=768,
                                 t_heads=12, t_layers=12, vocab=49408, embed=256, batch=2, gated=False, spread=True),
    'tsfb_224_f16_b2_spread': dict(img=224, patch=16, frames=16, dim=768, depth=12, heads=12, t_width=512,
                                   t_heads=8, t_layers=12, vocab=49408, embed=256, batch=2, gated=False, spread=True),
    # 9217 tokens per clip: the reference runs with use_checkpoint=True (its own activation checkpointing of the two
    # attention modules, timesformer.py:175-187: same values, ~1/2 of the 60+ GB the plain backward would keep)
    'tsfl14_336_f16_b2_spread': dict(img=336, patch=14, frames=16, dim=1024, depth=24, heads=16, t_width=768,
                                     t_heads=12, t_layers=12, vocab=49408, embed=256, batch=2, gated=False, spread=True,
                                     checkpoint=True),
}


def build_reference_model(ref, c):
    vis = ref.timesformer.SpaceTimeTransformer(
        img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'],
        num_heads=c['heads'], num_frames=c['frames'], time_init='zeros',
        attention_style='frozen-in-time', ln_pre=True, act_layer=ref.openai_model.QuickGELU,
        is_tanh_gating=c['gated'])
    vis.head = nn.Identity()
    vis.pre_logits = nn.Identity()
    vis.fc = nn.Identity()
    model = ref.models.CLIP(
        embed_dim=c['embed'], vision_width=c['dim'], vision_model=vis, context_length=77,
        vocab_size=c['vocab'], transformer_width=c['t_width'], transformer_heads=c['t_heads'],
        transformer_layers=c['t_layers'], tempearture_init=0.07)
    return model


def synthetic_inputs(c, seed=1234):
    video, tokens = O.synthetic_batch(c['batch'], c['frames'], c['img'], seed=seed, spread=c.get('spread', False))
    if c['vocab'] < 49408:      # tiny vocab: remap ids, keep EOT as the maximum id
        tokens = tokens.clone()
        body = tokens[:, 1:31] % (c['vocab'] - 2) + 1
        tokens[:, 1:31] = body
        tokens[:, 0] = c['vocab'] - 2
        tokens[:, 31] = c['vocab'] - 1
    return video, tokens


def select_gradients(grads, c):
    """Format-2 fixtures: which gradients are stored in full. Every tensor outside the block stacks that is small
    (cls / positional / temporal embeddings, the LayerNorms, logit_scale), every 1-D gradient (LayerNorm gains and
    biases, Linear biases) of three video blocks and three text blocks (first, middle, last), and -- `slices` -- four
    rows (0, 1, a middle one, the last) of every weight gradient of those blocks and of the two projections, the patch
    embedding and the token table's used rows: direction information for the big tensors at a few KB each."""
    vb = sorted({0, c['depth'] // 2, c['depth'] - 1})
    tb = sorted({0, c['t_layers'] // 2, c['t_layers'] - 1})
    pref = [f'visual.blocks.{i}.' for i in vb] + [f'transformer.resblocks.{i}.' for i in tb]
    full, slices = {}, {}
    for k, g in grads.items():
        in_blocks = k.startswith('visual.blocks.') or k.startswith('transformer.resblocks.')
        chosen = any(k.startswith(p) for p in pref)
        if g.ndim <= 1 or k in ('visual.cls_token', 'visual.temporal_embed'):
            if not in_blocks or chosen:
                full[k] = g.clone()
        elif chosen or k in ('image_projection', 'text_projection', 'visual.patch_embed.proj.weight', 'visual.pos_embed',
                             'positional_embedding'):
            g2 = g.reshape(-1, g.shape[-1]) if k in ('visual.pos_embed', 'positional_embedding') else g.reshape(g.shape[0], -1)
            rows = sorted({0, 1, g2.shape[0] // 2, g2.shape[0] - 1})
            slices[k] = (rows, g2[rows].clone())
    return full, slices


def run_model_golden(ref, name, c):
    torch.manual_seed(0)
    model = build_reference_model(ref, c)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    weights = O.procedural_weights(shapes, seed=7, spread=c.get('spread', False))
    model.load_state_dict(weights, strict=True)
    model.train()
    video, tokens = synthetic_inputs(c)

    acts = {}

    def hook(tag):
        def fn(mod, inp, out):
            acts[tag] = out.detach().clone()
        return fn
    hs = [model.visual.blocks[0].timeattn.register_forward_hook(hook('blk0_timeattn_out')),
          model.visual.blocks[0].attn.register_forward_hook(hook('blk0_spaceattn_out')),
          model.visual.blocks[0].register_forward_hook(hook('blk0_out')),
          model.visual.blocks[-1].register_forward_hook(hook('blk_last_out')),
          model.transformer.resblocks[0].register_forward_hook(hook('txt_blk0_out_LND'))]

    out = model(video, tokens, use_checkpoint=bool(c.get('checkpoint', False)), norm_embed=True)
    crit = ref.loss.CLIPLoss(use_vissl=False, cache_labels=True, rank=0, world_size=1)
    ld = crit(out)
    ld['loss'].backward()
    for h in hs:
        h.remove()
    li = (out['logit_scale'] * out['image_embed'] @ out['text_embed'].T).detach()

    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    fixture = {
        'config': c, 'shapes': shapes, 'weight_seed': 7, 'input_seed': 1234,
        'image_embed': out['image_embed'].detach(), 'text_embed': out['text_embed'].detach(),
        'logit_scale': out['logit_scale'].detach(), 'logits_per_image': li,
        'loss': ld['loss'].detach(), 'clip_acc': ld['clip_acc'].detach(),
        'pred': li.argmax(-1), 'labels': torch.arange(li.shape[0]),
        'state_dict_keys': list(model.state_dict().keys()),
        'param_names': [k for k, _ in model.named_parameters()],
    }
    if c['dim'] <= 128:
        fixture['acts'] = acts
        fixture['grads'] = grads
        with torch.no_grad():       # narrator-style call: all tokens, not only the cls row (timesformer.py:377-381)
            fixture['features_all_tokens'] = model.visual.forward_features(
                video.permute(0, 2, 1, 3, 4).contiguous(), use_checkpoint=False, cls_at_last=False).detach().clone()
    elif c.get('spread'):      # fixture format 2 (round 5)
        fixture['format'] = 2
        fixture['grad_norms'] = {k: g.norm().item() for k, g in grads.items()}
        fixture['grad_absmax'] = {k: g.abs().max().item() for k, g in grads.items()}
        full, slices = select_gradients(grads, c)
        fixture['grads'] = full
        fixture['grad_slices'] = slices
        fixture['acts'] = {k: v[:, :3].clone() if v.ndim == 3 else v for k, v in acts.items()}
        e_i, e_t = out['image_embed'].detach(), out['text_embed'].detach()
        off = ~torch.eye(e_i.shape[0], dtype=torch.bool)
        fixture['sample_cosines'] = {'image_mean': (e_i @ e_i.T)[off].mean().item(), 'image_max': (e_i @ e_i.T)[off].max().item(),
                                     'text_mean': (e_t @ e_t.T)[off].mean().item(), 'text_max': (e_t @ e_t.T)[off].max().item()}
        print(f'[golden] {name}: {len(full)} full gradients, {len(slices)} weight-gradient slices, sample cosines '
              f'{fixture["sample_cosines"]}, pred {fixture["pred"].tolist()}')
    else:      # full-size model: keep per-parameter grad norms + a few full small grads
        fixture['grad_norms'] = {k: g.norm().item() for k, g in grads.items()}
        keep = ['logit_scale', 'visual.cls_token', 'visual.temporal_embed', 'visual.norm.weight',
                'visual.blocks.0.norm3.weight', 'visual.blocks.11.timeattn.qkv.bias',
                'visual.blocks.5.attn.proj.bias', 'ln_final.bias',
                'transformer.resblocks.3.attn.in_proj_bias', 'visual.ln_pre.weight']
        fixture['grads'] = {k: grads[k] for k in keep}
        fixture['acts'] = {k: v[:, :3].clone() if v.ndim == 3 else v for k, v in acts.items()}
    torch.save(fixture, os.path.join(GOLDEN, f'model_{name}.pt'))
    print(f'[golden] {name}: loss={ld["loss"].item():.6f} acc={ld["clip_acc"].item():.1f} '
          f'params={sum(v.numel() for v in weights.values())/1e6:.2f}M')


def run_var_attention_golden(ref):
    """VarAttention (timesformer.py:87-144) on odd shapes, both einops modes, with grads."""
    cases = []
    for (B, Fr, N, H) in [(2, 3, 5, 2), (2, 4, 49, 2), (3, 1, 7, 2), (2, 16, 4, 3)]:
        D = 64 * H
        torch.manual_seed(100 + B + Fr + N)
        m = ref.timesformer.VarAttention(D, num_heads=H, qkv_bias=True)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        w = O.procedural_weights(shapes, seed=11)
        m.load_state_dict(w)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, 1 + Fr * N, D, generator=g)
        gout = torch.randn(B, 1 + Fr * N, D, generator=g)
        rec = {'B': B, 'F': Fr, 'N': N, 'H': H, 'shapes': shapes, 'x': x, 'gout': gout}
        for mode, (efrom, eto, dims) in {
            'space': ('b (f n) d', '(b f) n d', {'f': Fr}),
            'time': ('b (f n) d', '(b n) f d', {'n': N}),
        }.items():
            xi = x.clone().requires_grad_(True)
            m.zero_grad()
            y = m(xi, efrom, eto, dims)
            y.backward(gout)
            rec[mode] = {'y': y.detach(), 'dx': xi.grad.clone(),
                         'dw': {k: p.grad.clone() for k, p in m.named_parameters()}}
        cases.append(rec)
    torch.save(cases, os.path.join(GOLDEN, 'var_attention.pt'))
    print(f'[golden] var_attention: {len(cases)} cases')


def _rank_worker(rank, world, port, use_vissl, q, local_loss=False, gather_with_grad=False):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ref = load_reference()
    g = torch.Generator().manual_seed(77)
    E, Bl = 16, 3
    img = O.l2_normalize(torch.randn(world * Bl, E, generator=g))
    txt = O.l2_normalize(torch.randn(world * Bl, E, generator=g))
    li = img[rank * Bl:(rank + 1) * Bl].clone().requires_grad_(True)
    lt = txt[rank * Bl:(rank + 1) * Bl].clone().requires_grad_(True)
    scale = torch.tensor(14.285714).requires_grad_(True)
    crit = ref.loss.CLIPLoss(use_vissl=use_vissl, local_loss=local_loss, gather_with_grad=gather_with_grad,
                             cache_labels=True, rank=rank, world_size=world)
    out = crit({'image_embed': li, 'text_embed': lt, 'logit_scale': scale})
    out['loss'].backward()
    q.put((rank, out['loss'].item(), out['clip_acc'].item(), li.grad.tolist(), lt.grad.tolist(),
           scale.grad.item()))
    dist.barrier()
    dist.destroy_process_group()


def _gather_rank_worker(rank, world, port, with_grad, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ref = load_reference()
    g = torch.Generator().manual_seed(55)
    E, Bl = 8, 3
    img = torch.randn(world * Bl, E, generator=g)
    txt = torch.randn(world * Bl, E, generator=g)
    wa = torch.randn(world, world * Bl, E, generator=g)       # a different downstream weighting on every rank
    wb = torch.randn(world, world * Bl, E, generator=g)
    li = img[rank * Bl:(rank + 1) * Bl].clone().requires_grad_(True)
    lt = txt[rank * Bl:(rank + 1) * Bl].clone().requires_grad_(True)
    ai, at = ref.loss.gather_features(li, lt, local_loss=False, gather_with_grad=with_grad, rank=rank, world_size=world)
    ((ai * wa[rank]).sum() + (at * wb[rank]).sum()).backward()
    q.put((rank, ai.detach().tolist(), at.detach().tolist(), li.grad.tolist(), lt.grad.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def run_gather_features_golden():
    """gather_features (loss.py:18-43) on 2 gloo ranks, with and without gather_with_grad."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    results = {}
    port = 29671
    world = 2
    for with_grad in (False, True):
        q = ctx.Queue()
        procs = [ctx.Process(target=_gather_rank_worker, args=(r, world, port, with_grad, q)) for r in range(world)]
        port += 1
        for p in procs:
            p.start()
        got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join()
        results[with_grad] = {'all_img': [torch.tensor(g[1]) for g in got], 'all_txt': [torch.tensor(g[2]) for g in got],
                              'dimg': [torch.tensor(g[3]) for g in got], 'dtxt': [torch.tensor(g[4]) for g in got]}
        print(f'[golden] gather_features with_grad={with_grad}')
    torch.save({'E': 8, 'B_local': 3, 'seed': 55, 'world': world, 'results': results},
               os.path.join(GOLDEN, 'gather_features.pt'))


def run_multirank_loss_golden():
    """CLIPLoss on 2 and 3 gloo ranks, vissl and non-vissl (loss.py:69-118, distributed_utils.py:51-89)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    results = {}
    port = 29611
    for world in (2, 3):
        for use_vissl in (True, False):
            q = ctx.Queue()
            procs = [ctx.Process(target=_rank_worker, args=(r, world, port, use_vissl, q)) for r in range(world)]
            port += 1
            for p in procs:
                p.start()
            got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
            for p in procs:
                p.join()
            results[(world, use_vissl)] = {
                'loss': [g[1] for g in got], 'acc': [g[2] for g in got],
                'dimg': torch.cat([torch.tensor(g[3]) for g in got]),
                'dtxt': torch.cat([torch.tensor(g[4]) for g in got]),
                'dscale': [g[5] for g in got]}
            print(f'[golden] multirank world={world} vissl={use_vissl} loss={got[0][1]:.6f}')
    # local_loss=True (loss.py:86-88, 99-100): every rank keeps its own loss; with / without gather_with_grad
    for world, with_grad in ((2, False), (2, True), (3, False)):
        q = ctx.Queue()
        procs = [ctx.Process(target=_rank_worker, args=(r, world, port, False, q, True, with_grad)) for r in range(world)]
        port += 1
        for p in procs:
            p.start()
        got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join()
        results[(world, 'local', with_grad)] = {
            'loss': [g[1] for g in got], 'acc': [g[2] for g in got],
            'dimg': torch.cat([torch.tensor(g[3]) for g in got]),
            'dtxt': torch.cat([torch.tensor(g[4]) for g in got]),
            'dscale': [g[5] for g in got]}
        print(f'[golden] multirank local_loss world={world} gather_with_grad={with_grad} losses={[round(g[1], 5) for g in got]}')
    torch.save({'E': 16, 'B_local': 3, 'seed': 77, 'scale': 14.285714, 'results': results},
               os.path.join(GOLDEN, 'clip_loss_multirank.pt'))


_ssl_inputs = O.ssl_synthetic_inputs


def _ssl_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ref = load_reference()
    E, Bl = 16, 4
    img, txt, ind = _ssl_inputs(world * Bl, E, 91)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    li, lt = img[sl].clone().requires_grad_(True), txt[sl].clone().requires_grad_(True)
    scale = torch.tensor(14.285714).requires_grad_(True)
    crit = ref.loss.SSLCLIPLoss(use_vissl=True, cache_labels=True, rank=rank, world_size=world, scale_init=0.08)
    out = crit({'image_embed': li, 'text_embed': lt, 'logit_scale': scale}, ind[sl].clone())
    out['loss'].backward()
    q.put((rank, {k: float(v) for k, v in out.items()}, li.grad.tolist(), lt.grad.tolist(), scale.grad.item(),
           crit.logit_scale_pseudo.grad.item()))
    dist.barrier()
    dist.destroy_process_group()


def run_ssl_loss_golden(ref):
    """SSLCLIPLoss (loss.py:121-217): single process, and 2 gloo ranks with use_vissl=True."""
    import torch.multiprocessing as mp
    img, txt, ind = _ssl_inputs(12, 16, 91)
    li, lt = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
    scale = torch.tensor(14.285714).requires_grad_(True)
    crit = ref.loss.SSLCLIPLoss(use_vissl=False, cache_labels=True, rank=0, world_size=1, scale_init=0.08)
    out = crit({'image_embed': li, 'text_embed': lt, 'logit_scale': scale}, ind.clone())
    out['loss'].backward()
    single = {'out': {k: float(v) for k, v in out.items()}, 'dimg': li.grad.clone(), 'dtxt': lt.grad.clone(),
              'dscale': scale.grad.item(), 'dpseudo_param': crit.logit_scale_pseudo.grad.item()}
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_ssl_rank_worker, args=(r, world, 29655, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join()
    multi = {'world': world, 'B_local': 4, 'out': [g[1] for g in got],
             'dimg': torch.cat([torch.tensor(g[2]) for g in got]), 'dtxt': torch.cat([torch.tensor(g[3]) for g in got]),
             'dscale': [g[4] for g in got], 'dpseudo_param': [g[5] for g in got]}
    torch.save({'E': 16, 'seed': 91, 'scale': 14.285714, 'scale_init': 0.08, 'single_G': 12, 'single': single,
                'multi': multi}, os.path.join(GOLDEN, 'ssl_clip_loss.pt'))
    print(f"[golden] ssl_clip_loss: single loss={single['out']['loss']:.6f} multi loss={multi['out'][0]['loss']:.6f}")


NARRATOR = dict(img=32, patch=16, frames=2, dim=128, depth=2, heads=2, text_width=192, queries=24, pool_heads=3, batch=3)


def run_narrator_pool_golden(ref):
    """Tower-side seam of the narrator, from the reference's own modules: SpaceTimeTransformer.forward_features(
    cls_at_last=False) -> coca.CrossAttention(norm_context=True) on the repeated img_queries -> coca.LayerNorm, glued
    exactly as VCLM_HF.encode_image does (narrator.py:63-90; narrator.py itself does not import on this container's
    transformers, its 12 glue lines are restated here around the reference's modules)."""
    c = NARRATOR
    torch.manual_seed(0)
    vis = ref.timesformer.SpaceTimeTransformer(
        img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'], num_heads=c['heads'],
        num_frames=c['frames'], time_init='zeros', attention_style='frozen-in-time', ln_pre=True,
        act_layer=ref.openai_model.QuickGELU, is_tanh_gating=False)
    vis.head = nn.Identity()
    vis.pre_logits = nn.Identity()
    vis.fc = nn.Identity()

    class Seam(nn.Module):                       # attribute names of VCLM_HF (narrator.py:44-49)
        def __init__(self):
            super().__init__()
            self.visual = vis
            self.img_queries = nn.Parameter(torch.empty(c['queries'], c['text_width']))
            self.img_attn_pool = ref.coca.CrossAttention(dim=c['text_width'], context_dim=c['dim'], dim_head=64,
                                                         heads=c['pool_heads'], norm_context=True)
            self.img_attn_pool_norm = ref.coca.LayerNorm(c['text_width'])

        def encode_image(self, image):           # narrator.py:73-90, SpaceTimeTransformer branch
            image = image.permute(0, 2, 1, 3, 4).contiguous()
            x = self.visual.forward_features(image, use_checkpoint=False, cls_at_last=False)
            x = x.permute(0, 2, 1)
            x = x.flatten(start_dim=2)
            x = x.permute(0, 2, 1)
            q = self.img_queries[None].expand(x.shape[0], -1, -1)
            q = self.img_attn_pool(q, x)
            return self.img_attn_pool_norm(q), x
    seam = Seam()
    shapes = {k: tuple(v.shape) for k, v in seam.state_dict().items()}
    weights = O.procedural_weights(shapes, seed=13)
    for k in shapes:
        if k.endswith('.beta'):                  # the zero buffers stay zero (coca.py:31)
            weights[k] = torch.zeros(shapes[k])
    seam.load_state_dict(weights, strict=True)
    seam.eval()
    video, _ = O.synthetic_batch(c['batch'], c['frames'], c['img'], seed=77)
    with torch.no_grad():
        tokens, feats = seam.encode_image(video)
        g = torch.Generator().manual_seed(5)
        xq = torch.randn(2, 10, c['text_width'], generator=g)          # per-sample queries: the general module call
        ctx = torch.randn(2, 37, c['dim'], generator=g)
        pool_general = seam.img_attn_pool(xq, ctx)
    torch.save({'config': c, 'shapes': shapes, 'weight_seed': 13, 'input_seed': 77, 'state_dict_keys': list(shapes),
                'image_tokens': tokens, 'features': feats, 'pool_general': pool_general, 'pool_general_seed': 5},
               os.path.join(GOLDEN, 'narrator_pool.pt'))
    print('narrator_pool', tuple(tokens.shape), float(tokens.abs().mean()))


DECODER = dict(vocab=331, positions=40, layers=3, text_len=13, max_text_length=15,
               variants={'freq1_gated': dict(cross_attn_freq=1, gated_xattn=True),
                         'freq2_plain': dict(cross_attn_freq=2, gated_xattn=False)})


def decoder_weights(model, seed):
    """Procedural weights for every learnable tensor of a VCLM_HF; the causal-mask buffers (`attn.bias`,
    `attn.masked_bias`, gpt2_gated.py:154-160) and the coca `beta` zeros keep their constructed values and lm_head
    stays tied to wte (transformers 4.27 ties them in post_init; `tie_word_embeddings` is GPT-2's default)."""
    sd = model.state_dict()
    keep = [k for k in sd if k.endswith('.attn.bias') or k.endswith('.crossattention.bias') or k.endswith('masked_bias')
            or k.endswith('.beta')]
    shapes = {k: tuple(v.shape) for k, v in sd.items() if k not in keep and k != 'text_decoder.lm_head.weight'}
    weights = O.procedural_weights(shapes, seed=seed)
    weights['text_decoder.lm_head.weight'] = weights['text_decoder.transformer.wte.weight']
    for k in keep:
        weights[k] = sd[k].clone()
    return shapes, keep, weights


def run_narrator_decoder_golden():
    """The narrator end to end from the reference's own, unmodified narrator.py / gpt2_gated.py / coca.py /
    timesformer.py (imported under transformers-4.27 name stand-ins, oracle/ref_import.py): VCLM_HF.forward (teacher-forced
    logits), VCLM_HF.generate with top_k=1 (free-running, early-stopping, and teacher-forced with a target)."""
    from transformers import GPT2Config
    from oracle.ref_import import load_reference_narrator
    ref = load_reference_narrator()
    c, d = NARRATOR, DECODER
    out = {'config': c, 'decoder': {k: v for k, v in d.items() if k != 'variants'}, 'variants': {}}
    for vi, (name, var) in enumerate(d['variants'].items()):
        torch.manual_seed(0)
        vis = ref.timesformer.SpaceTimeTransformer(
            img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'], num_heads=c['heads'],
            num_frames=c['frames'], time_init='zeros', attention_style='frozen-in-time', ln_pre=True,
            act_layer=ref.openai_model.QuickGELU, is_tanh_gating=False)
        vis.head = nn.Identity()
        vis.pre_logits = nn.Identity()
        vis.fc = nn.Identity()
        base = GPT2Config(vocab_size=d['vocab'], n_positions=d['positions'], n_embd=c['text_width'], n_layer=d['layers'],
                          n_head=c['pool_heads'], use_cache=False, bos_token_id=d['vocab'] - 1, eos_token_id=d['vocab'] - 1)
        cfg = ref.gpt2_gated.augment_gpt2_config(base, **var)
        dec = ref.gpt2_gated.GPT2LMHeadModel(cfg)
        model = ref.narrator.VCLM_HF(vision_width=c['dim'], vision_model=vis, text_width=c['text_width'],
                                     text_decoder=dec, num_img_queries=c['queries'], dim_head=64, heads=c['pool_heads'])
        shapes, keep, weights = decoder_weights(model, seed=29 + vi)
        model.load_state_dict(weights, strict=True)
        dec.lm_head.weight = dec.transformer.wte.weight
        model.eval()
        video, _ = O.synthetic_batch(c['batch'], c['frames'], c['img'], seed=78)
        g = torch.Generator().manual_seed(11 + vi)
        bos = d['vocab'] - 1
        text = torch.randint(1, d['vocab'] - 1, (c['batch'], d['text_len']), generator=g)
        text[:, 0] = bos
        text[1, 9:] = 0                                              # pad (id 0) tail: ignored by the target nll
        with torch.no_grad():
            fwd = model(video, text)
            image_tokens = model.encode_image(video)
            tok = types.SimpleNamespace(bos_token_id=bos, eos_token_id=-1, pad_token_id=0)
            free_ids, free_ppl = model.generate(image_tokens, tok, max_text_length=d['max_text_length'], top_k=1)
            # an eos id that the free-running greedy decode really emits (row 0, step 5), so the eos bookkeeping is live
            tok.eos_token_id = int(free_ids[0, 6])
            eos_ids, eos_ppl = model.generate(image_tokens, tok, max_text_length=d['max_text_length'], top_k=1)
            stop_ids, stop_ppl = model.generate(image_tokens[:1], tok, max_text_length=d['max_text_length'], top_k=1,
                                                early_stopping=True)
            tf_ids, tf_ppl = model.generate(image_tokens, tok, target=text, max_text_length=d['text_len'], top_k=1,
                                            teacher_forcing=True)
            tgt_ids, tgt_ppl = model.generate(image_tokens, tok, target=text, max_text_length=d['text_len'], top_k=1)
            rep_ids, rep_ppl = model.generate(image_tokens, tok, max_text_length=8, top_k=1, num_return_sequences=2)
        out['variants'][name] = {
            'variant': var, 'shapes': shapes, 'kept_buffers': keep, 'weight_seed': 29 + vi, 'input_seed': 78,
            'text': text, 'bos': bos, 'eos': tok.eos_token_id, 'pad': 0,
            'image_tokens': image_tokens, 'logits': fwd['text_tokens_logits'], 'labels': fwd['labels'],
            'free_ids': free_ids, 'free_ppl': free_ppl, 'eos_ids': eos_ids, 'eos_ppl': eos_ppl,
            'stop_ids': stop_ids, 'stop_ppl': stop_ppl, 'tf_ids': tf_ids, 'tf_ppl': tf_ppl,
            'tgt_ids': tgt_ids, 'tgt_ppl': tgt_ppl, 'rep_ids': rep_ids, 'rep_ppl': rep_ppl,
        }
        print('narrator_decoder', name, tuple(fwd['text_tokens_logits'].shape), float(fwd['text_tokens_logits'].abs().mean()),
              free_ids[0].tolist(), free_ppl.tolist(), tuple(stop_ids.shape))
    torch.save(out, os.path.join(GOLDEN, 'narrator_decoder.pt'))


def run_narrator_beam_golden():
    """`VCLM_HF.beam_sample` and `VCLM_HF.group_beam_search` of the unmodified narrator.py (narrator.py:149-366) on the
    narrator_decoder.pt models (same procedural weights, same image tokens), with the transformers-4.27 BeamSearchScorer
    restated in oracle/beam_scorer.py (the installed transformers has none). beam_sample draws 2 * num_beams candidates
    with torch.multinomial: with top_k = 2 (min_tokens_to_keep = 2 for beams, narrator.py:380-383) every beam keeps exactly
    two tokens, the draw without replacement returns ALL 2 * num_beams candidates and the following sort makes the
    outcome independent of the random stream -- the deterministic setting the goldens use."""
    from transformers import GPT2Config
    from oracle.ref_import import load_reference_narrator
    ref = load_reference_narrator()
    c, d = NARRATOR, DECODER
    out = {'variants': {}}
    for vi, (name, var) in enumerate(d['variants'].items()):
        torch.manual_seed(0)
        vis = ref.timesformer.SpaceTimeTransformer(
            img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'], num_heads=c['heads'],
            num_frames=c['frames'], time_init='zeros', attention_style='frozen-in-time', ln_pre=True,
            act_layer=ref.openai_model.QuickGELU, is_tanh_gating=False)
        vis.head = nn.Identity()
        vis.pre_logits = nn.Identity()
        vis.fc = nn.Identity()
        base = GPT2Config(vocab_size=d['vocab'], n_positions=d['positions'], n_embd=c['text_width'], n_layer=d['layers'],
                          n_head=c['pool_heads'], use_cache=False, bos_token_id=d['vocab'] - 1, eos_token_id=d['vocab'] - 1)
        dec = ref.gpt2_gated.GPT2LMHeadModel(ref.gpt2_gated.augment_gpt2_config(base, **var))
        model = ref.narrator.VCLM_HF(vision_width=c['dim'], vision_model=vis, text_width=c['text_width'],
                                     text_decoder=dec, num_img_queries=c['queries'], dim_head=64, heads=c['pool_heads'])
        shapes, keep, weights = decoder_weights(model, seed=29 + vi)
        model.load_state_dict(weights, strict=True)
        dec.lm_head.weight = dec.transformer.wte.weight
        model.eval()
        video, _ = O.synthetic_batch(c['batch'], c['frames'], c['img'], seed=78)
        bos = d['vocab'] - 1
        runs = {}
        with torch.no_grad():
            image_tokens = model.encode_image(video)
            tok = types.SimpleNamespace(bos_token_id=bos, eos_token_id=-1, pad_token_id=0)
            L = 12
            ids0, _ = model.group_beam_search(image_tokens, tok, max_text_length=L, num_beams=4, num_beam_groups=2)
            # eos ids that the searches really emit, so that hypotheses close and entries finish at different steps
            eos_a, eos_b = int(ids0[0, 3]), int(ids0[1, 5])
            cases = {
                'gbs_free': ('group_beam_search', dict(num_beams=6, num_beam_groups=3, num_return_sequences=2), -1),
                'gbs_eos': ('group_beam_search', dict(num_beams=4, num_beam_groups=2, num_return_sequences=1), eos_a),
                'gbs_eos_lp': ('group_beam_search', dict(num_beams=6, num_beam_groups=2, num_return_sequences=3,
                                                         length_penalty=0.6, top_k=5, temperature=0.8), eos_b),
                'gbs_one_group': ('group_beam_search', dict(num_beams=3, num_beam_groups=1, num_return_sequences=2,
                                                            length_penalty=1.5), eos_a),
                'bs_free': ('beam_sample', dict(num_beams=3, top_k=2), -1),
                'bs_eos': ('beam_sample', dict(num_beams=3, top_k=2, temperature=0.7, length_penalty=0.8), eos_b),
                'bs_rep': ('beam_sample', dict(num_beams=2, top_k=2, num_return_sequences=2), eos_a),
            }
            for cname, (fn, kw, eos) in cases.items():
                tok = types.SimpleNamespace(bos_token_id=bos, eos_token_id=eos, pad_token_id=0)
                torch.manual_seed(5)
                seq, score = getattr(model, fn)(image_tokens, tok, max_text_length=L, **kw)
                runs[cname] = {'fn': fn, 'kwargs': kw, 'eos': eos, 'max_text_length': L, 'sequences': seq,
                               'sequence_scores': score}
                print('narrator_beam', name, cname, tuple(seq.shape), seq[0].tolist(), [round(x, 4) for x in score.tolist()])
        out['variants'][name] = {'bos': bos, 'pad': 0, 'runs': runs}
    torch.save(out, os.path.join(GOLDEN, 'narrator_beam.pt'))


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    ref = load_reference()
    if not only or 'attn' in only:
        run_var_attention_golden(ref)
    for name, c in CONFIGS.items():
        if not only or 'model' in only or f'model:{name}' in only:
            run_model_golden(ref, name, c)
    if not only or 'gather' in only:
        run_gather_features_golden()
    if not only or 'multirank' in only:
        run_multirank_loss_golden()
    if not only or 'ssl' in only:
        run_ssl_loss_golden(ref)
    if not only or 'narrator' in only:
        run_narrator_pool_golden(ref)
    if not only or 'decoder' in only:
        run_narrator_decoder_golden()
    if not only or 'beam' in only:
        run_narrator_beam_golden()


if __name__ == '__main__':
    main()
