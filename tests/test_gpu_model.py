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


@pytest.mark.parametrize('name', ['tiny_p16', 'tiny_p14_gated', 'config1_tsfb_112'])
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
