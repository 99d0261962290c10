"""CPU: pins oracle/oracle.py (the restatement) against the committed outputs of the real
reference (tests/golden/*.pt, made by oracle/gen_golden.py). Tolerances are fp32 round-off."""
import pytest
import torch

from oracle import oracle as O
from oracle.gen_golden import synthetic_inputs
from conftest import load_golden
from helpers import check_fixture_gradients, fixture_weights

import os

MODELS = ['tiny_p16', 'tiny_p14_gated', 'tiny_f16', 'config1_tsfb_112', 'config2_tsfb_224_b8_spread']
# the TSF-L/14 and 16-frame fixtures (24 blocks of width 1024; 3137 / 9217 tokens per clip) take minutes of CPU and tens
# of GB each: checked on request (LAVILA_SLOW_ORACLE=1; done when the fixtures were generated)
if os.environ.get('LAVILA_SLOW_ORACLE') == '1':
    MODELS += ['tsfl14_224_b2_spread', 'tsfl14_336_b2_spread', 'tsfb_224_f16_b2_spread']
if os.environ.get('LAVILA_SLOW_ORACLE') == '2':
    MODELS += ['tsfl14_336_f16_b2_spread']


@pytest.mark.parametrize('case', range(4))
@pytest.mark.parametrize('mode', ['space', 'time'])
def test_var_attention_matches_reference(case, mode):
    rec = load_golden('var_attention.pt')[case]
    w = O.procedural_weights(rec['shapes'], seed=11)
    x = rec['x'].clone().requires_grad_(True)
    ws = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    y = O.var_attention(x, ws, '', rec['H'], rec['F'], rec['N'], mode)
    y.backward(rec['gout'])
    torch.testing.assert_close(y.detach(), rec[mode]['y'], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(x.grad, rec[mode]['dx'], atol=2e-5, rtol=1e-4)
    for k, g in rec[mode]['dw'].items():
        torch.testing.assert_close(ws[k].grad, g, atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize('name', MODELS)
def test_full_model_matches_reference(name):
    fx = load_golden(f'model_{name}.pt')
    c = fx['config']
    # 9217 tokens per clip x 24 blocks: the oracle's plain backward would keep ~80 GB -- forward pins only (the backward
    # of that shape is pinned on the GPU against the reference's gradients, tests/test_gpu_f32_class.py)
    fwd_only = bool(c.get('checkpoint'))
    w = {k: v.requires_grad_(v.is_floating_point() and not fwd_only) for k, v in fixture_weights(fx).items()}
    video, tokens = synthetic_inputs(c, seed=fx['input_seed'])
    with torch.set_grad_enabled(not fwd_only):
        out = O.clip_forward(video, tokens, w, c['heads'], c['t_heads'], norm_embed=True)
    # format-2 ("spread") fixtures: attention scores of a few units amplify float32 round-off ~3x (two float32
    # evaluation orders of the same network: 1.1e-5 on a text embedding entry)
    ea = 3e-5 if fx.get('format', 1) == 2 else 1e-5
    torch.testing.assert_close(out['image_embed'], fx['image_embed'], atol=ea, rtol=1e-4)
    torch.testing.assert_close(out['text_embed'], fx['text_embed'], atol=ea, rtol=1e-4)
    ld = O.clip_loss(out['image_embed'], out['text_embed'], out['logit_scale'])
    torch.testing.assert_close(ld['logits_per_image'], fx['logits_per_image'], atol=1e-4 * ea / 1e-5, rtol=1e-4)
    torch.testing.assert_close(ld['loss'], fx['loss'], atol=1e-5, rtol=1e-5)
    assert torch.equal(ld['labels'], fx['labels'])          # int64, bit-exact
    assert torch.equal(ld['pred'], fx['pred'])
    torch.testing.assert_close(ld['clip_acc'], fx['clip_acc'])
    if fwd_only:
        return
    ld['loss'].backward()
    if fx.get('format', 1) == 2:
        check_fixture_gradients(fx, {k: v.grad for k, v in w.items() if v.grad is not None}, rtol=1e-3, norm_rtol=2e-3)
        return
    for k, g in fx['grads'].items():
        torch.testing.assert_close(w[k].grad, g, atol=2e-5, rtol=2e-3, msg=lambda m: f'{k}: {m}')
    if 'grad_norms' in fx:
        for k, n in fx['grad_norms'].items():
            assert abs(w[k].grad.norm().item() - n) <= 2e-3 * n + 1e-7, k


@pytest.mark.parametrize('name', ['tiny_p16', 'tiny_f16'])
def test_all_token_features_match_reference(name):
    """forward_features(cls_at_last=False) (the narrator's call, timesformer.py:377-381): every token row of the
    final LayerNorm, not only the cls row."""
    fx = load_golden(f'model_{name}.pt')
    c = fx['config']
    w = O.procedural_weights(fx['shapes'], seed=fx['weight_seed'])
    video, _ = synthetic_inputs(c, seed=fx['input_seed'])
    feats = O.vision_tower(video, w, c['heads'], cls_at_last=False)
    assert feats.shape == (c['batch'], 1 + c['frames'] * (c['img'] // c['patch']) ** 2, c['dim'])
    torch.testing.assert_close(feats, fx['features_all_tokens'], atol=2e-5, rtol=1e-4)


def test_multirank_loss_equals_single_process_on_concatenation():
    """every rank's CLIPLoss equals the single-process loss on the rank-ordered
    concatenation; vissl local grads = W x the global-loss gradient slice, non-vissl = 1 x."""
    fx = load_golden('clip_loss_multirank.pt')
    for key, r in fx['results'].items():
        if len(key) != 2:
            continue                        # local_loss cases: next test
        world, use_vissl = key
        g = torch.Generator().manual_seed(fx['seed'])
        G = world * fx['B_local']
        img = O.l2_normalize(torch.randn(G, fx['E'], generator=g)).requires_grad_(True)
        txt = O.l2_normalize(torch.randn(G, fx['E'], generator=g)).requires_grad_(True)
        scale = torch.tensor(fx['scale']).requires_grad_(True)
        ld = O.clip_loss(img, txt, scale)
        ld['loss'].backward()
        for rank in range(world):
            assert abs(r['loss'][rank] - ld['loss'].item()) < 1e-6
            assert abs(r['acc'][rank] - ld['clip_acc'].item()) < 1e-4
            assert abs(r['dscale'][rank] - scale.grad.item()) < 1e-6
        mult = world if use_vissl else 1
        torch.testing.assert_close(r['dimg'], mult * img.grad, atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(r['dtxt'], mult * txt.grad, atol=1e-6, rtol=1e-5)


def test_multirank_local_loss_is_the_mean_over_the_rank_rows():
    """CLIPLoss(local_loss=True) (loss.py:86-88, 99-100): rank r's loss = mean over ITS rows of the two cross-entropies
    against all gathered partners. Without gather_with_grad the partners are constants (row terms only); with it every
    rank's loss reaches every embedding, i.e. the local gradient is that of the SUM of the per-rank losses."""
    import torch.nn.functional as F
    fx = load_golden('clip_loss_multirank.pt')
    seen = 0
    for key, r in fx['results'].items():
        if len(key) != 3:
            continue
        world, _, with_grad = key
        seen += 1
        g = torch.Generator().manual_seed(fx['seed'])
        Bl, G = fx['B_local'], world * fx['B_local']
        img = O.l2_normalize(torch.randn(G, fx['E'], generator=g))
        txt = O.l2_normalize(torch.randn(G, fx['E'], generator=g))
        dimg, dtxt = torch.zeros_like(img), torch.zeros_like(txt)
        for rank in range(world):
            sl = slice(rank * Bl, (rank + 1) * Bl)
            ia, ta = img.clone().requires_grad_(True), txt.clone().requires_grad_(True)
            scale = torch.tensor(fx['scale']).requires_grad_(True)
            il = ia[sl] if with_grad else img[sl].clone().requires_grad_(True)
            tl = ta[sl] if with_grad else txt[sl].clone().requires_grad_(True)
            labels = torch.arange(rank * Bl, (rank + 1) * Bl)
            li, lt = scale * il @ (ta if with_grad else txt).t(), scale * tl @ (ia if with_grad else img).t()
            loss = (F.cross_entropy(li, labels) + F.cross_entropy(lt, labels)) / 2
            loss.backward()
            assert abs(r['loss'][rank] - loss.item()) < 1e-6
            assert abs(r['acc'][rank] - 100.0 * (li.argmax(-1) == labels).float().mean().item()) < 1e-4
            assert abs(r['dscale'][rank] - scale.grad.item()) < 1e-6
            if with_grad:
                dimg += ia.grad
                dtxt += ta.grad
            else:
                dimg[sl], dtxt[sl] = il.grad, tl.grad
        torch.testing.assert_close(r['dimg'], dimg, atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(r['dtxt'], dtxt, atol=1e-6, rtol=1e-5)
    assert seen == 3


def test_ssl_clip_loss_matches_reference():
    """SSLCLIPLoss (loss.py:121-217): single process and the 2-rank vissl run (every rank = loss on the
    rank-ordered concatenation; local grads = W x the global-loss slice; scale grads = full derivative)."""
    import math
    fx = load_golden('ssl_clip_loss.pt')
    for G, want, mult in ((fx['single_G'], fx['single'], 1), (fx['multi']['world'] * fx['multi']['B_local'], fx['multi'], fx['multi']['world'])):
        img, txt, ind = O.ssl_synthetic_inputs(G, fx['E'], fx['seed'])
        img.requires_grad_(True), txt.requires_grad_(True)
        scale = torch.tensor(fx['scale']).requires_grad_(True)
        pparam = torch.tensor(math.log(1 / fx['scale_init'])).requires_grad_(True)
        ld = O.ssl_clip_loss(img, txt, ind, scale, pparam.exp())
        ld['loss'].backward()
        outs = want['out'] if isinstance(want['out'], list) else [want['out']]
        for o in outs:
            for k in ('loss', 'clip_acc', 'clip_acc_gt', 'clip_acc_pseudo', 'num_gt', 'num_pseudo'):
                assert abs(o[k] - float(ld[k])) < 1e-4, (k, o[k], float(ld[k]))
        torch.testing.assert_close(want['dimg'], mult * img.grad, atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(want['dtxt'], mult * txt.grad, atol=1e-6, rtol=1e-5)
        ds = want['dscale'] if isinstance(want['dscale'], list) else [want['dscale']]
        dp = want['dpseudo_param'] if isinstance(want['dpseudo_param'], list) else [want['dpseudo_param']]
        for a in ds:
            assert abs(a - scale.grad.item()) < 1e-6
        for a in dp:
            assert abs(a - pparam.grad.item()) < 1e-6


def test_narrator_pool_oracle_matches_reference():
    """oracle.narrator_encode_image / cross_attention_pool against the reference's own coca.CrossAttention + LayerNorm on
    top of its SpaceTimeTransformer.forward_features(cls_at_last=False) (tests/golden/narrator_pool.pt)."""
    fx = load_golden('narrator_pool.pt')
    c = fx['config']
    w = O.procedural_weights(fx['shapes'], seed=fx['weight_seed'])
    for k in fx['shapes']:
        if k.endswith('.beta'):
            w[k] = torch.zeros(fx['shapes'][k])
    video, _ = O.synthetic_batch(c['batch'], c['frames'], c['img'], seed=fx['input_seed'])
    with torch.no_grad():
        feats = O.vision_tower(video, w, c['heads'], cls_at_last=False)
        tokens = O.narrator_encode_image(video, w, c['heads'], c['pool_heads'])
        g = torch.Generator().manual_seed(fx['pool_general_seed'])
        xq = torch.randn(2, 10, c['text_width'], generator=g)
        ctx = torch.randn(2, 37, c['dim'], generator=g)
        general = O.cross_attention_pool(xq, ctx, w, 'img_attn_pool.', c['pool_heads'])
    torch.testing.assert_close(feats, fx['features'], atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(tokens, fx['image_tokens'], atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(general, fx['pool_general'], atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize('variant', ['freq1_gated', 'freq2_plain'])
def test_narrator_decoder_oracle_matches_reference(variant):
    """oracle.narrator_forward / narrator_generate_greedy against the reference's own VCLM_HF.forward and
    VCLM_HF.generate(top_k=1) (unmodified narrator.py + gpt2_gated.py, tests/golden/narrator_decoder.pt): teacher-forced
    logits, free-running ids / perplexities with and without an eos, early stopping, target scoring with and without
    teacher forcing, num_return_sequences -- each with the reference's full-prefix recompute AND with a key/value cache."""
    fx = load_golden('narrator_decoder.pt')
    c, d, v = fx['config'], fx['decoder'], fx['variants'][variant]
    w = O.narrator_weights(v['shapes'], seed=v['weight_seed'])
    video, _ = O.synthetic_batch(c['batch'], c['frames'], c['img'], seed=v['input_seed'])
    H = c['pool_heads']
    with torch.no_grad():
        out = O.narrator_forward(video, v['text'], w, c['heads'], H, H)
        torch.testing.assert_close(out['text_tokens_logits'], v['logits'], atol=2e-4, rtol=1e-4)
        assert torch.equal(out['labels'], v['labels'])
        img = O.narrator_encode_image(video, w, c['heads'], H)
        torch.testing.assert_close(img, v['image_tokens'], atol=1e-5, rtol=1e-4)
        for cache in (False, True):
            kw = dict(w=w, dec_heads=H, bos=v['bos'], pad=v['pad'], use_cache=cache)
            ids, ppl = O.narrator_generate_greedy(img, eos=-1, max_text_length=d['max_text_length'], **kw)
            assert torch.equal(ids, v['free_ids'])
            torch.testing.assert_close(ppl, v['free_ppl'], atol=0, rtol=2e-4)
            ids, ppl = O.narrator_generate_greedy(img, eos=v['eos'], max_text_length=d['max_text_length'], **kw)
            assert torch.equal(ids, v['eos_ids'])
            torch.testing.assert_close(ppl, v['eos_ppl'], atol=0, rtol=2e-4)
            ids, ppl = O.narrator_generate_greedy(img[:1], eos=v['eos'], max_text_length=d['max_text_length'],
                                                  early_stopping=True, **kw)
            assert torch.equal(ids, v['stop_ids'])
            torch.testing.assert_close(ppl, v['stop_ppl'], atol=0, rtol=2e-4)
            ids, ppl = O.narrator_generate_greedy(img, eos=v['eos'], max_text_length=d['text_len'], target=v['text'],
                                                  teacher_forcing=True, **kw)
            assert torch.equal(ids, v['tf_ids'])
            torch.testing.assert_close(ppl, v['tf_ppl'], atol=0, rtol=1e-3)
            ids, ppl = O.narrator_generate_greedy(img, eos=v['eos'], max_text_length=d['text_len'], target=v['text'], **kw)
            assert torch.equal(ids, v['tgt_ids'])
            torch.testing.assert_close(ppl, v['tgt_ppl'], atol=0, rtol=1e-3)
            ids, ppl = O.narrator_generate_greedy(img.repeat_interleave(2, dim=0), eos=v['eos'], max_text_length=8, **kw)
            assert torch.equal(ids, v['rep_ids'])
            torch.testing.assert_close(ppl, v['rep_ppl'], atol=0, rtol=2e-4)


def test_bench_generator_equals_the_oracles():
    """bench.py carries its own synthetic-batch generator (the product bench does not import oracle/ for its inputs);
    it must produce exactly the batches the parity tests draw from oracle.synthetic_batch."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(os.path.dirname(__file__), '..', 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for args in [(3, 2, 32, 7), (2, 4, 48, 1234)]:
        v0, t0 = O.synthetic_batch(*args[:3], seed=args[3])
        v1, t1 = bench.synthetic_batch(*args[:3], seed=args[3])
        assert torch.equal(v0, v1) and torch.equal(t0, t1)
