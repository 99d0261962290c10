"""GPU (-m gpu): the whole dual encoder (lavila_amd behind the reference import paths) against the committed
outputs of the reference (tests/golden/model_*.pt): embeddings, logits, loss within 1e-3 (float32), labels and
argmax bit-exact, gradients of every parameter."""
import pytest
import torch

from conftest import load_golden
from helpers import build_model
from oracle import oracle as O
from oracle.gen_golden import synthetic_inputs

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _run(name, norm_embed=True):
    from lavila.models.loss import CLIPLoss
    fx = load_golden(f'model_{name}.pt')
    c = fx['config']
    model = build_model(c)
    model.load_state_dict(O.procedural_weights(fx['shapes'], seed=fx['weight_seed']), strict=True)
    model.to(DEV).train()
    video, tokens = synthetic_inputs(c, seed=fx['input_seed'])
    out = model(video.to(DEV), tokens.to(DEV), norm_embed=norm_embed)
    crit = CLIPLoss(use_vissl=False, cache_labels=True, rank=0, world_size=1)
    ld = crit(out)
    ld['loss'].backward()
    return fx, model, out, ld, crit


@pytest.mark.parametrize('name', ['tiny_p16', 'tiny_p14_gated', 'tiny_f16', 'config1_tsfb_112'])
def test_model_matches_reference_fp32(name):
    fx, model, out, ld, crit = _run(name)
    assert set(out) == {'image_embed', 'text_embed', 'logit_scale'}
    assert set(ld) == {'loss', 'clip_loss', 'clip_acc'} and ld['loss'].ndim == 0
    tol = dict(atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(out['image_embed'].cpu(), fx['image_embed'], **tol)
    torch.testing.assert_close(out['text_embed'].cpu(), fx['text_embed'], **tol)
    torch.testing.assert_close(out['logit_scale'].detach().cpu(), fx['logit_scale'], **tol)
    dbg = crit.debug_slabs(out)
    torch.testing.assert_close(dbg['logits'][0].cpu(), fx['logits_per_image'], **tol)
    torch.testing.assert_close(dbg['logits'][1].cpu(), fx['logits_per_image'].t(), **tol)
    assert torch.equal(dbg['labels'].cpu(), fx['labels'])             # int64 label indices: bit-exact
    assert torch.equal(dbg['pred'][0].cpu(), fx['pred'])               # argmax indices: bit-exact
    torch.testing.assert_close(ld['loss'].detach().cpu(), fx['loss'], **tol)
    torch.testing.assert_close(ld['clip_acc'].cpu(), fx['clip_acc'])
    grads = {k: p.grad for k, p in model.named_parameters()}
    assert all(g is not None for g in grads.values())
    for k, g in fx['grads'].items():
        torch.testing.assert_close(grads[k].cpu(), g, atol=1e-4, rtol=5e-3, msg=lambda m: f'{k}: {m}')
    for k, n in fx.get('grad_norms', {}).items():
        got = grads[k].norm().item()
        assert abs(got - n) <= 5e-3 * n + 1e-6, (k, got, n)


def test_intermediate_activations_match_reference():
    from lavila.models.timesformer import SpaceTimeBlock
    fx = load_golden('model_tiny_p16.pt')
    c = fx['config']
    model = build_model(c)
    model.load_state_dict(O.procedural_weights(fx['shapes'], seed=fx['weight_seed']), strict=True)
    model.to(DEV).eval()
    video, tokens = synthetic_inputs(c, seed=fx['input_seed'])
    vis = model.visual
    with torch.no_grad():
        tok = vis.patch_embed.tokens_from_bcthw(video.to(DEV))
        from lavila_amd import ops
        n = vis.patches_per_frame
        x = vis.ln_pre(ops.embed_tokens(tok, vis.cls_token, vis.pos_embed, vis.temporal_embed, c['frames'], n))
        blk = vis.blocks[0]
        t_out = blk.timeattn(blk.norm3(x), 'b (f n) d', '(b n) f d', {'n': n})
        torch.testing.assert_close(t_out.cpu(), fx['acts']['blk0_timeattn_out'], atol=1e-4, rtol=1e-3)
        y = blk(x, 'b (f n) d', '(b f) n d', 'b (f n) d', '(b n) f d', time_n=n, space_f=c['frames'])
        torch.testing.assert_close(y.cpu(), fx['acts']['blk0_out'], atol=2e-4, rtol=1e-3)
        # text block 0 through the reference (LND) signature
        xt = (model.token_embedding(tokens.to(DEV)) + model.positional_embedding).permute(1, 0, 2)
        yt = model.transformer.resblocks[0](xt)
        torch.testing.assert_close(yt.cpu(), fx['acts']['txt_blk0_out_LND'], atol=2e-4, rtol=1e-3)


def test_eval_api_encode_and_checkpoint_flag():
    """eval_zeroshot.py calls encode_image / encode_text (eval_zeroshot.py:312,322); use_checkpoint=True must
    give the same numbers (activation checkpointing is a memory knob)."""
    fx = load_golden('model_tiny_p16.pt')
    c = fx['config']
    model = build_model(c)
    model.load_state_dict(O.procedural_weights(fx['shapes'], seed=fx['weight_seed']), strict=True)
    model.to(DEV).eval()
    video, tokens = synthetic_inputs(c, seed=fx['input_seed'])
    with torch.no_grad():
        ie = model.encode_image(video.to(DEV))
        te = model.encode_text(tokens.to(DEV))
        feat = model.encode_image(video.to(DEV), apply_project=False)
    assert feat.shape == (c['batch'], c['dim'])
    torch.testing.assert_close(O.l2_normalize(ie.cpu()), fx['image_embed'], atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(O.l2_normalize(te.cpu()), fx['text_embed'], atol=1e-3, rtol=1e-3)
    model.train()
    out = model(video.to(DEV), tokens.to(DEV), use_checkpoint=True, norm_embed=True)
    (out['image_embed'].sum() + out['text_embed'].sum()).backward()
    torch.testing.assert_close(out['image_embed'].detach().cpu(), fx['image_embed'], atol=1e-3, rtol=1e-3)
    assert model.visual.blocks[0].timeattn.qkv.weight.grad is not None


def test_activation_checkpointing_on_the_benched_path_changes_nothing():
    """main_pretrain.py:100,491-495 (`--use-checkpoint`; timesformer.py:175-187): activation checkpointing per block is a
    memory knob -- BASELINE configs[2] at its real per-GPU shape (TSF-B, 16 x 224^2, local batch 256) needs it to fit in
    288 GB. At a geometry where every feature of the benched bf16 path is on (width 768: own GEMMs with residual epilogues,
    pending MLPs across block boundaries, column-sum tokens, bias-gradient riders, cls-only last block; two towers on two
    streams), loss and every parameter gradient must equal the plain run's up to bf16 rounding, and two checkpointed runs
    must agree to the bit (the step is deterministic since round 6). Round 6 found this path broken: a backward read ctx.saved_tensors twice, which torch.utils.checkpoint's unpack hook
    refuses (CheckpointError at the first real use, profiles/r06_bench_config3_b256_16f.json)."""
    from lavila.models.loss import CLIPLoss
    cfg = dict(img=224, patch=16, frames=4, dim=768, depth=3, heads=12, t_width=512, t_heads=8, t_layers=2, vocab=512,
               embed=256, batch=2, gated=False)
    video, tokens = O.synthetic_batch(cfg['batch'], cfg['frames'], cfg['img'], seed=7)
    tokens = tokens.clone()
    tokens[:, 1:20] = tokens[:, 1:20] % 510 + 1
    tokens[:, 0], tokens[:, 20] = 510, 511
    tokens[:, 21:] = 0

    def run(ckpt):
        model = build_model(cfg)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(O.procedural_weights(shapes, seed=5))
        model.to(DEV).train()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss = CLIPLoss()(model(video.to(DEV), tokens.to(DEV), use_checkpoint=ckpt, norm_embed=True))['loss']
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    l0, g0 = run(False)
    l1, g1 = run(True)
    # Not bit-identical: under checkpointing a block materialises its pending MLP at the block boundary (the residual sum
    # is rounded to bf16 once more on its way into the saved tensor) -- bf16-rounding differences, nothing structural
    assert abs(l0 - l1) < 2e-3 * max(1.0, abs(l0)), (l0, l1)
    assert set(g0) == set(g1) and len(g0) > 40
    worst = max(((g0[n].float() - g1[n].float()).norm() / (g0[n].float().norm() + 1e-12)).item() for n in g0)
    assert worst < 5e-2, worst
    assert all(bool(torch.isfinite(g).all()) for g in g1.values())
    l2, g2 = run(True)                       # and the checkpointed step is itself reproducible to the bit
    assert l1 == l2 and all(torch.equal(g1[n], g2[n]) for n in g1)


def test_bf16_autocast_training_step_tracks_fp32():
    """The perf path: bf16 autocast (and fp16 autocast remapped to bf16). Loose tolerance: bf16 activations."""
    fx = load_golden('model_tiny_p16.pt')
    c = fx['config']
    for amp_dtype in (torch.bfloat16, torch.float16):
        model = build_model(c)
        model.load_state_dict(O.procedural_weights(fx['shapes'], seed=fx['weight_seed']), strict=True)
        model.to(DEV).train()
        video, tokens = synthetic_inputs(c, seed=fx['input_seed'])
        from lavila.models.loss import CLIPLoss
        with torch.autocast('cuda', dtype=amp_dtype):
            out = model(video.to(DEV), tokens.to(DEV), norm_embed=True)
            ld = CLIPLoss()(out)
        ld['loss'].backward()
        torch.testing.assert_close(out['image_embed'].float().cpu(), fx['image_embed'], atol=4e-2, rtol=4e-2)
        torch.testing.assert_close(out['text_embed'].float().cpu(), fx['text_embed'], atol=4e-2, rtol=4e-2)
        assert abs(ld['loss'].item() - fx['loss'].item()) < 0.1
        g = model.visual.blocks[1].mlp.fc1.weight.grad
        assert g is not None and torch.isfinite(g).all()
        ref = fx['grads']['visual.blocks.1.mlp.fc1.weight']
        cos = torch.nn.functional.cosine_similarity(g.float().cpu().flatten(), ref.flatten(), dim=0)
        assert cos > 0.98, cos


@pytest.mark.parametrize('name', ['tiny_p16', 'tiny_f16'])
@pytest.mark.parametrize('amp', [False, True])
def test_all_token_features_cls_at_last_false(name, amp):
    """forward_features(x, cls_at_last=False) -- the narrator's call (narrator.py:74, timesformer.py:377-381): the final
    LayerNorm on EVERY token row (the default path normalises the B cls rows only), float32 and bf16 (F=16: the
    register-tiled 16-frame time kernels)."""
    fx = load_golden(f'model_{name}.pt')
    c = fx['config']
    model = build_model(c)
    model.load_state_dict(O.procedural_weights(fx['shapes'], seed=fx['weight_seed']), strict=True)
    model.to(DEV).eval()
    video, _ = synthetic_inputs(c, seed=fx['input_seed'])
    x = video.to(DEV).permute(0, 2, 1, 3, 4).contiguous()              # reference signature: [B, F, C, H, W]
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        feats = model.visual.forward_features(x, use_checkpoint=False, cls_at_last=False)
    want = fx['features_all_tokens']
    assert feats.shape == want.shape
    if amp:
        # 2 blocks of bf16 roundings on O(1) LayerNorm outputs: eps*sqrt(10*2) ~ 5e-3 relative, bound 3e-2 absolute
        torch.testing.assert_close(feats.float().cpu(), want, atol=3e-2, rtol=3e-2)
    else:
        torch.testing.assert_close(feats.cpu(), want, atol=2e-4, rtol=1e-3)


def test_grad_scaler_step_is_harmless():
    """main_pretrain.py:490-521 wraps the step in fp16 autocast + GradScaler (initial scale 65536). fp16 autocast is
    re-entered as bf16; the scaled loss must flow through every hand-written backward (the contrastive backward takes
    the upstream gradient as a DEVICE scalar) and scaler.step() must apply the same update as an unscaled step."""
    from lavila.models.loss import CLIPLoss
    fx = load_golden('model_tiny_p16.pt')
    c = fx['config']
    video, tokens = synthetic_inputs(c, seed=fx['input_seed'])
    results = []
    for scaled in (False, True):
        model = build_model(c)
        model.load_state_dict(O.procedural_weights(fx['shapes'], seed=fx['weight_seed']), strict=True)
        model.to(DEV).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        scaler = torch.amp.GradScaler('cuda', enabled=scaled)
        with torch.autocast('cuda', dtype=torch.float16):
            out = model(video.to(DEV), tokens.to(DEV), norm_embed=True)
            loss = CLIPLoss()(out)['loss']
        scaler.scale(loss).backward()
        if scaled:
            assert scaler.get_scale() == 65536.0
            g = model.visual.blocks[0].attn.qkv.weight.grad
            assert torch.isfinite(g).all() and g.abs().max() > 1.0      # gradients really carry the 2^16 factor
        scaler.step(opt)
        scaler.update()
        assert not scaled or scaler.get_scale() == 65536.0            # no inf/nan was found: the scale is kept
        results.append({k: p.detach().clone() for k, p in model.named_parameters()})
    for k in results[0]:
        torch.testing.assert_close(results[1][k], results[0][k], atol=2e-4, rtol=2e-2, msg=lambda m, k=k: f'{k}: {m}')


def test_block_forward_backward_is_hip_graph_capturable():
    """include/lavila_hip.h promises that every entry point is stream-ordered, allocation-free and capturable: one
    SpaceTimeBlock (LayerNorm / GEMM / attention / fused MLP kernels, forward AND backward) is captured into a
    hipGraph and replayed on new data; the replay must equal eager execution bit for bit."""
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeBlock
    torch.manual_seed(0)
    Fr, N, D, H, B = 4, 196, 768, 12, 2
    blk = SpaceTimeBlock(D, H, qkv_bias=True, act_layer=QuickGELU, time_init='rand').to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            if p.ndim > 1:
                p.normal_(0, 0.02)
    x_static = torch.randn(B, 1 + Fr * N, D, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    g_static = torch.randn(B, 1 + Fr * N, D, device=DEV, dtype=torch.bfloat16)

    def step():
        for p in blk.parameters():
            p.grad = None
        x_static.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16):
            x1, y, b = blk.chain(x_static, None, None, Fr, N)
            out = x1 + y + b.to(y.dtype)
        out.backward(g_static)
        return out

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):             # warm-up outside capture (lazy attribute setting, weight copies)
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_static = step()
    gx_static = x_static.grad
    gw_static = blk.mlp.fc1.weight.grad
    # new data into the static buffers, replay, compare with eager on the same data
    with torch.no_grad():
        x_static.copy_(torch.randn_like(x_static))
        g_static.copy_(torch.randn_like(g_static))
    graph.replay()
    torch.cuda.synchronize()
    got = (out_static.clone(), gx_static.clone(), gw_static.clone())
    out_e = step()
    torch.cuda.synchronize()
    assert torch.equal(got[0], out_e)
    # input gradient: the cls token's d(q / k / v) of the space attention are summed over the frames with float32
    # atomics (attn_space_bwd.hip), whose order is not fixed. A last-bit difference there reaches the cls row directly
    # and -- through the cls query of the TIME attention, which attends to every token (timesformer.py:116-119) -- the
    # dK / dV of any token, where it occasionally flips one bf16 rounding. So: every row within one bf16 ulp-scale
    # tolerance, and all but a handful of patch rows bit-equal (observed: 0 rows in five runs out of six, 4 of 1570 in
    # the sixth; demanding bit equality made this test fail about one run in five).
    bad = (got[1] != x_static.grad).any(-1).nonzero().tolist()
    patch_bad = [bt for bt in bad if bt[1] != 0]
    assert len(patch_bad) <= 16, f'input gradient differs on {len(patch_bad)} patch rows: {patch_bad[:8]}'
    torch.testing.assert_close(got[1].float(), x_static.grad.float(), atol=2e-2, rtol=2e-2)
    nbad = int((got[2] != blk.mlp.fc1.weight.grad).sum())
    assert nbad == 0, f'fc1 weight gradient differs in {nbad} elements'

