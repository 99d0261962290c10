"""GPU (-m gpu): lavila_amd.graph_step.GraphedTrainStep -- the pretraining iteration (main_pretrain.py:482-530: forward,
CLIPLoss, backward, AdamW, logit-scale clamp) captured as one hipGraph per caption-length bucket.

Two copies of one model see the same five batches: one through the eager loop (per-batch caption read-back, float
learning rate written into the parameter groups), one through the graphed step (eager first call, then capture / replay,
bucketed caption bound taken from the HOST tokens, learning rate as a device scalar). Every launch is the same kernel on
the same operands, so losses and gradients agree to rounding noise of the float32 atomics (cls rows of the attention
backward), and the replays really are replays (two graphs for two buckets, host time of a replay far below a step's)."""
import copy

import pytest
import torch

from helpers import build_model

pytestmark = pytest.mark.gpu
DEV = 'cuda'

CFG = dict(img=224, patch=16, frames=4, dim=768, depth=2, heads=12, t_width=512, t_heads=8, t_layers=2, vocab=1024,
           embed=256, batch=4, gated=False)
LENGTHS = [20, 23, 33, 20, 40]              # caption lengths (EOT position + 1): buckets of 8 -> 24, 24, 40, 24, 40
LRS = [1e-4, 2e-4, 3e-4, 2e-4, 1e-4]


def _batches():
    g = torch.Generator().manual_seed(5)
    out = []
    for n in LENGTHS:
        video = torch.randn(CFG['batch'], 3, CFG['frames'], CFG['img'], CFG['img'], generator=g)
        tokens = torch.zeros(CFG['batch'], 77, dtype=torch.long)
        for b in range(CFG['batch']):
            ln = n if b == 0 else max(3, n - 2 * b)             # the first caption is the longest
            tokens[b, :ln - 1] = torch.randint(1, CFG['vocab'] - 2, (ln - 1,), generator=g)
            tokens[b, ln - 1] = CFG['vocab'] - 1                  # EOT = highest id
        out.append((video, tokens))
    return out


def _groups(model):
    decay = [p for n, p in model.named_parameters() if p.ndim >= 2]
    rest = [p for n, p in model.named_parameters() if p.ndim < 2]
    return [{'params': decay, 'weight_decay': 0.01}, {'params': rest, 'weight_decay': 0.0}]


def test_graphed_step_equals_the_eager_loop():
    from lavila.models.loss import CLIPLoss
    from lavila_amd.graph_step import GraphedTrainStep
    torch.manual_seed(3)
    model_e = build_model(CFG).to(DEV).train()
    model_g = copy.deepcopy(model_e)
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    # eps far above the rounding noise of a gradient: with the default 1e-8 Adam's first steps move every parameter by
    # +-lr whatever its gradient's size, and the float32 atomics' noise on near-zero gradients would decide signs -- two
    # EAGER runs then already drift apart by 2 lr per step on those elements (seen: 3 % gradient difference at step 2)
    opt_e = torch.optim.AdamW(_groups(model_e), lr=1e-4, eps=1e-3, fused=True, capturable=True)
    opt_g = torch.optim.AdamW(_groups(model_g), lr=1e-4, eps=1e-3, fused=True, capturable=True)
    step = GraphedTrainStep(model_g, crit, opt_g, (CFG['batch'], 3, CFG['frames'], CFG['img'], CFG['img']),
                            (CFG['batch'], 77), DEV)
    losses_e, losses_g, grads_e, grads_g = [], [], [], []
    for it, (video, tokens) in enumerate(_batches()):
        # eager loop, as main_pretrain.py writes it
        for grp in opt_e.param_groups:
            grp['lr'] = LRS[it]
        v, t = video.to(DEV), tokens.to(DEV)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            le = crit(model_e(v, t, use_checkpoint=False, norm_embed=True))['loss']
        le.backward()
        grads_e.append(torch.cat([p.grad.flatten().float() for p in model_e.parameters()]).clone())
        opt_e.step()
        opt_e.zero_grad(set_to_none=True)
        model_e.logit_scale.data.clamp_(0, 4.6052)
        losses_e.append(float(le))
        # graphed step on the HOST tensors
        step.set_lr(LRS[it])
        out = step(video, tokens)
        losses_g.append(float(out['loss']))
        if it > 0:                       # graph-owned gradients of the replay that just ran
            grads_g.append(torch.cat([p.grad.flatten().float() for p in model_g.parameters()]).clone())
        else:
            grads_g.append(None)
    assert step.buckets == [24, 40] and step.replays == 4
    for it in range(len(LENGTHS)):
        # atomics noise of the cls rows, amplified by four Adam steps: 2.2e-3 relative seen at step 4 on one box
        assert abs(losses_e[it] - losses_g[it]) <= 4e-3 * abs(losses_e[it]) + 1e-4, (it, losses_e, losses_g)
    assert losses_e[-1] != losses_e[0]
    # step 1 (first replay) starts from states that differ by atomics noise at most: gradients agree in aggregate
    for it in (1, 2):
        ge, gg = grads_e[it], grads_g[it]
        rel = float((ge - gg).norm() / ge.norm())
        assert rel < 3e-2, (it, rel)
    # parameters after five steps: Adam turns a rounding-noise gradient into a +-lr move, so compare in aggregate
    pe = torch.cat([p.detach().flatten().float() for p in model_e.parameters()])
    pg = torch.cat([p.detach().flatten().float() for p in model_g.parameters()])
    moved = float((pe - pg).abs().max())
    assert moved <= 2.0 * sum(LRS) + 1e-6, moved
    frac = float(((pe - pg).abs() > 0.25 * max(LRS)).float().mean())
    assert frac < 0.02, frac


def test_replay_is_host_light_and_refuses_what_it_cannot_replay():
    import time
    from lavila.models.loss import CLIPLoss
    from lavila_amd.graph_step import GraphedTrainStep
    torch.manual_seed(4)
    model = build_model(CFG).to(DEV).train()
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    with pytest.raises(ValueError, match='capturable'):
        GraphedTrainStep(model, crit, torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True), (1,), (1, 77), DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True, capturable=True)
    shape_v, shape_t = (CFG['batch'], 3, CFG['frames'], CFG['img'], CFG['img']), (CFG['batch'], 77)
    step = GraphedTrainStep(model, crit, opt, shape_v, shape_t, DEV)
    video, tokens = _batches()[0]
    with pytest.raises(ValueError, match='built for'):
        step(video[:2], tokens[:2])
    for _ in range(3):
        step(video, tokens)
    torch.cuda.synchronize()
    # a replay: the host returns long before the device has finished the step
    t0 = time.perf_counter()
    out = step(video, tokens)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    assert torch.isfinite(out['loss'])
    assert host < 0.5 * total or total < 5e-3, (host, total)
    # a device token tensor without a bound: the full context (no read-back), one more bucket
    step(video.to(DEV), tokens.to(DEV))
    assert step.buckets == [24, 77]
    step(video.to(DEV), tokens.to(DEV), text_len=21)
    assert step.buckets == [24, 77]



def test_evaluation_between_replays_sees_the_updated_weights():
    """a replay moves the parameters on the device without touching a version counter; ops.weight_copies
    caches the GEMMs' bf16 weight copies across no-grad forwards. replay -> eval -> replay -> eval: every eval must run
    on the CURRENT weights -- compared against the same forward with the cache bypassed (lvl_cast_transpose on the live
    parameters) -- and the two evals must differ (the step in between really moved the weights)."""
    from lavila.models.loss import CLIPLoss
    from lavila_amd import ops
    from lavila_amd.graph_step import GraphedTrainStep
    torch.manual_seed(6)
    model = build_model(CFG).to(DEV).train()
    crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
    opt = torch.optim.AdamW(model.parameters(), lr=5e-3, fused=True, capturable=True)
    step = GraphedTrainStep(model, crit, opt, (CFG['batch'], 3, CFG['frames'], CFG['img'], CFG['img']), (CFG['batch'], 77), DEV)
    with pytest.raises(NotImplementedError, match='use_checkpoint'):
        GraphedTrainStep(model, crit, opt, (1,), (1, 77), DEV, forward_kwargs=dict(use_checkpoint=True))
    video, tokens = _batches()[0]
    v, t = video.to(DEV), tokens.to(DEV)

    def evaluate():
        model.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            cached = model(v, t, norm_embed=True)['image_embed'].float().clone()
            ops.invalidate_weight_cache()                      # ground truth: copies re-cast from the live parameters
            fresh = model(v, t, norm_embed=True)['image_embed'].float().clone()
        model.train()
        return cached, fresh

    step(video, tokens)                      # eager first call
    step(video, tokens)                      # capture + first replay
    evals = []
    for _ in range(3):
        step(video, tokens)                  # replay: parameters move on the device
        torch.cuda.synchronize()
        cached, fresh = evaluate()
        assert torch.equal(cached, fresh), 'an eval between replays ran on stale bf16 weight copies'
        evals.append(fresh)
    assert step.replays >= 3
    assert not torch.equal(evals[0], evals[1]) and not torch.equal(evals[1], evals[2])


def _poison_free_device_memory(streams, dev):
    """NaN into the cached free blocks of the given streams' pools and into a fresh segment per size class."""
    torch.cuda.synchronize()
    for st in streams:
        with torch.cuda.stream(st):
            junk = [torch.full((n // 4,), float('nan'), device=dev) for n in
                    (256, 2048, 16384, 131072, 1 << 20, 1 << 22, 1 << 24, 1 << 26) for _ in range(12)]
            del junk
    torch.cuda.synchronize()


@pytest.mark.parametrize('geometry', ['small', 'tsfb', 'long_text'])
def test_replay_does_not_depend_on_free_device_memory(geometry):
    """Two runs of the same five steps (eager, capture, three replays); the second fills every free block of the allocator
    with NaN in front of EVERY call, the eager one included. A step may only read what it owns, and it is a function of its
    inputs: losses and final parameters must be IDENTICAL to the bit.
    (Round 5: the hipMemsetAsync nodes of a replayed graph took their fill pattern from memory that had been recycled --
    the 'zeroed' class-token accumulators of the attention backward came back as {0, NaN, 0, 0} repeated; csrc/common.h.
    Round 6: the "second outcome" this test saw at the TSF-B geometry about one run in seven -- losses 1.40463 / 1.57504
    instead of 1.40535 / 1.58009 -- had nothing to do with the poison: a clean run takes it as often. The time attention's
    backward added 25 partial records of the cls token's gradient with f32 atomics, in timing order, and one bf16 rounding
    of dqkv's cls row sat on the fence (tools/probe_second_outcome.py, profiles/r06_second_outcome.txt). The records are
    summed in slot order now: csrc/attn_space_bwd.hip cls_grad_finalize_kernel. The captured graph must hold no memset
    node either: the token embedding's backward is an own kernel, csrc/text_embed.hip.)"""
    from lavila.models.loss import CLIPLoss
    from lavila_amd.graph_step import GraphedTrainStep
    from oracle import oracle as O
    cfg = dict(img=32, patch=16, frames=2, dim=256, depth=2, heads=4, t_width=256, t_heads=4, t_layers=2, vocab=512,
               embed=64, batch=3, gated=False) if geometry == 'small' else \
        dict(img=224, patch=16, frames=4, dim=768, depth=1, heads=12, t_width=512, t_heads=8, t_layers=1, vocab=512,
             embed=256, batch=4, gated=False)
    eot = 31
    if geometry == 'long_text':
        # 48 captions x 72 positions = 3456 token rows: above 3072 torch's embedding backward sorts the indices (rocPRIM
        # radix sort -- which zeroes its histograms with hipMemsetAsync, i.e. memset NODES inside the replayed graph); the
        # benched shape (256 x 32 rows) takes that path
        cfg = dict(img=32, patch=16, frames=2, dim=256, depth=1, heads=4, t_width=256, t_heads=4, t_layers=1, vocab=512,
                   embed=256, batch=48, gated=False)
        eot = 70
    dev = torch.device('cuda', torch.cuda.current_device())

    def run(poison):
        model = build_model(cfg)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(O.procedural_weights(shapes, seed=5))
        model.to(dev).train()
        crit = CLIPLoss(use_vissl=True, cache_labels=True, rank=0, world_size=1)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-3, fused=True, capturable=True)
        B = cfg['batch']
        step = GraphedTrainStep(model, crit, opt, (B, 3, cfg['frames'], cfg['img'], cfg['img']), (B, 77), dev)
        losses = []
        for it in range(5):
            video, tokens = O.synthetic_batch(B, cfg['frames'], cfg['img'], seed=40 + it)
            tokens = tokens.clone()
            tokens[:, 1:eot] = tokens[:, 1:eot] % 510 + 1
            tokens[:, 0], tokens[:, eot] = 510, 511
            tokens[:, eot + 1:] = 0
            if poison:
                _poison_free_device_memory([torch.cuda.current_stream(), step._stream], dev)
            losses.append(float(step(video, tokens)['loss']))
        torch.cuda.synchronize()
        assert step.replays == 4
        nodes.update(step.node_types)
        return losses, torch.cat([p.detach().flatten().float() for p in model.parameters()]).cpu()

    nodes = {}
    clean_l, clean_p = run(False)
    dirty_l, dirty_p = run(True)
    assert all(l == l and abs(l) < 1e4 for l in dirty_l) and bool(torch.isfinite(dirty_p).all()), (clean_l, dirty_l)
    assert clean_l == dirty_l, (clean_l, dirty_l)
    assert torch.equal(clean_p, dirty_p), f'max |d parameter| {float((clean_p - dirty_p).abs().max()):.3e}'
    assert all(n.get('memset', 0) == 0 for n in nodes.values()), nodes
