"""CPU (-m "not gpu"): the HOST logic of the narrator's decoder side -- the order of Conv1Ds, gates, fused add+LayerNorm
hand-overs and the key/value-cache bookkeeping of lavila_amd.gpt2_gated, and VCLM_HF.generate's sampling bookkeeping --
with every device primitive (the C-ABI calls) replaced by its oracle formula on CPU tensors. What is checked is the
plan around the kernels, against the reference's own outputs (tests/golden/narrator_decoder.pt); the kernels themselves
are checked on the GPU (tests/test_gpu_narrator.py). Nothing here is a product path: lavila_amd has no CPU fallback."""
import contextlib
import io
import types

import pytest
import torch

from conftest import load_golden
from oracle import oracle as O


def _emulate_device_primitives(monkeypatch):
    from lavila_amd import _cabi as C
    from lavila_amd import gpt2_gated as G
    from lavila_amd import ops
    monkeypatch.setattr(C, 'require_device', lambda *a: None)

    def embed(self, ids, L, pos_dev=None):
        ids = ids.reshape(-1)
        p0 = 0 if pos_dev is None else int(pos_dev)
        return (self.wte[ids] + self.wpe[p0 + torch.arange(ids.shape[0]) % L]).to(self.dtype)

    def add_ln(self, res, y, gate, ln):
        if y is not None:
            res += (1.0 if gate is None else gate) * y
        return O.layer_norm(res, ln[0], ln[1], self.eps)

    def act(self, u, which):
        u.copy_(O.gelu_new(u) if which == C.ACT_GELU_NEW else O.sq_relu(u))
        return u

    def cross_attn(self, q, kv, qrep):
        D = self.D
        ctx = kv.shape[0]
        return O.gpt2_attention_core(q.reshape(ctx, qrep, D), kv[..., :D], kv[..., D:], self.heads, causal=False).reshape(-1, D)

    def causal_attention(qkv, heads, bias=None):
        D = qkv.shape[-1] // 3
        return O.gpt2_attention_core(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], heads, causal=True)

    def self_attention(self, i, qkv):
        D = self.pack.D
        p = int(self.pos)
        self.cache[i][:, p] = qkv[:, D:]
        kv = self.cache[i][:, :p + 1]
        return O.gpt2_attention_core(qkv[:, None, :D], kv[..., :D], kv[..., D:], self.pack.heads, causal=True)[:, 0]

    def linear_f32_rows(x2, w3, bias=None, act_code=None):      # the f32-class GEMM on identity "term images"
        y = torch.nn.functional.linear(x2, w3, bias)
        if act_code is not None:
            y = O.gelu_new(y) if act_code == C.ACT_GELU_NEW else O.sq_relu(y)
        return y

    monkeypatch.setattr(ops, 'split3', lambda x2, role, stack=False: x2)
    monkeypatch.setattr(ops, 'linear_f32_rows', linear_f32_rows)
    monkeypatch.setattr(G._Pack, 'embed', embed)
    monkeypatch.setattr(G._Pack, 'add_ln', add_ln)
    monkeypatch.setattr(G._Pack, 'act', act)
    monkeypatch.setattr(G._Pack, 'cross_attn', cross_attn)
    monkeypatch.setattr(G.DecodeSession, '_self_attention', self_attention)
    monkeypatch.setattr(ops, 'causal_attention', causal_attention)


def _model(variant):
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeTransformer
    from lavila_amd.gpt2_gated import GPT2LMHeadModel, augment_gpt2_config, gpt2_config
    from lavila_amd.narrator import VCLM_HF
    fx = load_golden('narrator_decoder.pt')
    c, d, v = fx['config'], fx['decoder'], fx['variants'][variant]
    with contextlib.redirect_stdout(io.StringIO()):
        vis = SpaceTimeTransformer(img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'],
                                   num_heads=c['heads'], num_frames=c['frames'], time_init='zeros',
                                   attention_style='frozen-in-time', ln_pre=True, act_layer=QuickGELU)
    vis.head = vis.pre_logits = vis.fc = torch.nn.Identity()
    base = gpt2_config('gpt2', vocab_size=d['vocab'], n_positions=d['positions'], n_embd=c['text_width'],
                       n_layer=d['layers'], n_head=c['pool_heads'])
    dec = GPT2LMHeadModel(augment_gpt2_config(base, **v['variant']))
    m = VCLM_HF(vision_width=c['dim'], vision_model=vis, text_width=c['text_width'], text_decoder=dec,
                num_img_queries=c['queries'], dim_head=64, heads=c['pool_heads'])
    w = O.narrator_weights(v['shapes'], seed=v['weight_seed'])
    own = m.state_dict()
    assert set(v['shapes']) | set(v['kept_buffers']) | {'text_decoder.lm_head.weight'} == set(own)
    for k in v['kept_buffers']:
        w[k] = own[k]
    m.load_state_dict(w, strict=True)
    assert dec.lm_head.weight is dec.transformer.wte.weight
    return m.eval(), c, d, v


@pytest.mark.parametrize('variant', ['freq1_gated', 'freq2_plain'])
def test_decoder_plan_and_generate_bookkeeping(variant, monkeypatch):
    _emulate_device_primitives(monkeypatch)
    m, c, d, v = _model(variant)
    img = v['image_tokens']
    text = v['text']
    tok = types.SimpleNamespace(bos_token_id=v['bos'], eos_token_id=v['eos'], pad_token_id=v['pad'])
    no_eos = types.SimpleNamespace(bos_token_id=v['bos'], eos_token_id=-1, pad_token_id=v['pad'])
    with torch.no_grad():
        out = m.text_decoder(text[:, :-1].contiguous(), encoder_hidden_states=img)
        torch.testing.assert_close(out.logits.permute(0, 2, 1), v['logits'], atol=2e-4, rtol=1e-4)
        assert out[0] is out.logits
        lab = m.text_decoder(text, encoder_hidden_states=img, labels=text)
        assert lab.loss is not None and lab[0] is lab.loss and torch.isfinite(lab.loss)
        runs = [
            ('free', dict(tokenizer=no_eos, max_text_length=d['max_text_length'])),
            ('eos', dict(tokenizer=tok, max_text_length=d['max_text_length'])),
            ('tf', dict(tokenizer=tok, target=text, max_text_length=d['text_len'], teacher_forcing=True)),
            ('tgt', dict(tokenizer=tok, target=text, max_text_length=d['text_len'])),
            ('rep', dict(tokenizer=tok, max_text_length=8, num_return_sequences=2)),
        ]
        for name, kw in runs:
            for cache in (True, False):
                ids, ppl = m.generate(img, top_k=1, kv_cache=cache, graph=False, **kw)
                assert torch.equal(ids, v[name + '_ids']), (name, cache)
                torch.testing.assert_close(ppl, v[name + '_ppl'], atol=0, rtol=1e-3)
        ids, ppl = m.generate(img[:1], tok, max_text_length=d['max_text_length'], top_k=1, early_stopping=True, graph=False)
        assert torch.equal(ids, v['stop_ids'])
        torch.testing.assert_close(ppl, v['stop_ppl'], atol=0, rtol=1e-3)
        # a target whose first token is not bos: the reference conditions on it from the second step on (narrator.py:140)
        odd = text.clone()
        odd[:, 0] = 5
        a = m.generate(img, tok, target=odd, max_text_length=6, top_k=1, teacher_forcing=True, graph=False)
        want = O.narrator_generate_greedy(img, O.narrator_weights(v['shapes'], seed=v['weight_seed']), c['pool_heads'],
                                          v['bos'], v['eos'], v['pad'], 6, target=odd, teacher_forcing=True, use_cache=False)
        assert torch.equal(a[0], want[0])
        torch.testing.assert_close(a[1], want[1], atol=0, rtol=1e-3)


def test_reference_state_dict_names_and_shapes():
    """Every key / shape / dtype of the reference's gated GPT2LMHeadModel (gpt2_gated.py, imported unmodified) exists in
    lavila_amd.gpt2_gated.GPT2LMHeadModel and vice versa, for both cross-attention layouts."""
    from oracle.ref_import import load_reference_narrator, reference_available
    if not reference_available():
        pytest.skip('reference not mounted')
    from transformers import GPT2Config
    from lavila_amd.gpt2_gated import GPT2LMHeadModel, augment_gpt2_config
    ref = load_reference_narrator()
    for freq, gated in ((1, True), (2, False), (3, True)):
        base = GPT2Config(vocab_size=97, n_positions=24, n_embd=128, n_layer=4, n_head=2, use_cache=False,
                          bos_token_id=96, eos_token_id=96)
        theirs = ref.gpt2_gated.GPT2LMHeadModel(ref.gpt2_gated.augment_gpt2_config(base, cross_attn_freq=freq,
                                                                                   gated_xattn=gated)).state_dict()
        ours = GPT2LMHeadModel(augment_gpt2_config(base, cross_attn_freq=freq, gated_xattn=gated)).state_dict()
        assert set(ours) == set(theirs)
        for k in ours:
            assert ours[k].shape == theirs[k].shape and ours[k].dtype == theirs[k].dtype, k
            if k.endswith('.bias') and ours[k].dtype == torch.uint8 or k.endswith('masked_bias'):
                assert torch.equal(ours[k], theirs[k]), k


def test_vclm_constructors_and_cpu_is_loud():
    import warnings
    from lavila.models import models
    from lavila_amd._cabi import HipExtensionError
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter('ignore')
        m = models.VCLM_OPENAI_TIMESFORMER_BASE_GPT2(gated_xattn=True, freeze_lm_vclm=True, num_frames=4)
    dec = m.text_decoder
    assert len(dec.transformer.h) == 12 and all(b.has_cross for b in dec.transformer.h)
    assert dec.transformer.h[0].alpha_cattn.requires_grad and not dec.transformer.h[0].attn.c_attn.weight.requires_grad
    assert m.img_queries.shape == (256, 768) and dec.lm_head.weight is dec.transformer.wte.weight
    with torch.no_grad(), pytest.raises(HipExtensionError):
        dec(torch.zeros(1, 3, dtype=torch.long))
    import types
    tok = types.SimpleNamespace(bos_token_id=50256, eos_token_id=50256, pad_token_id=50256)
    with torch.no_grad(), pytest.raises(HipExtensionError):          # beam search is built (round 4): on CPU it is as loud
        m.group_beam_search(torch.zeros(1, 256, 768), tok, max_text_length=4, num_beams=2, num_beam_groups=1)


@pytest.mark.parametrize('top_k,top_p,temperature', [(None, 0.95, 0.7), (50, None, 1.0), (40, 0.9, 0.8), (None, None, 1.3),
                                                     (1, None, 1.0), (5, 0.3, 2.0), (None, 0.999, 1.0)])
def test_warp_equals_transformers_logits_warpers(top_k, top_p, temperature):
    """VCLM_HF._warp restates the warper list narrator.py:368-389 builds for num_beams=1; checked here against
    transformers' own TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper (the classes the reference imports,
    narrator.py:17-24): same kept set, same values."""
    lp = pytest.importorskip('transformers.generation.logits_process')
    from lavila_amd.narrator import VCLM_HF
    g = torch.Generator().manual_seed(17)
    logits = 3 * torch.randn(6, 997, generator=g)
    warpers = lp.LogitsProcessorList()
    if temperature is not None and temperature != 1.0:
        warpers.append(lp.TemperatureLogitsWarper(temperature))
    if top_k:
        warpers.append(lp.TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1))
    if top_p is not None and top_p < 1.0:
        warpers.append(lp.TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1))
    want = warpers(torch.zeros(6, 1, dtype=torch.long), logits.clone())
    got = VCLM_HF._warp(logits.clone(), top_k, top_p, temperature)
    assert torch.equal(torch.isfinite(got), torch.isfinite(want))
    keep = torch.isfinite(want)
    torch.testing.assert_close(got[keep], want[keep], atol=0, rtol=1e-6)


def test_decode_sessions_are_reused_per_shape_and_dropped_on_weight_change(monkeypatch):
    """generate() is called once per batch by the captioning drivers: the second batch of the same shape re-binds the
    cached session (buffers + captured graph) instead of building one; a parameter update invalidates it. Host logic only
    (device primitives emulated)."""
    _emulate_device_primitives(monkeypatch)
    m, c, d, v = _model('freq1_gated')
    dec = m.text_decoder
    img = v['image_tokens']
    tok = types.SimpleNamespace(bos_token_id=v['bos'], eos_token_id=-1, pad_token_id=v['pad'])
    with torch.no_grad():
        a = m.generate(img, tok, max_text_length=d['max_text_length'], top_k=1, graph=False)
        first = next(iter(dec._sessions.values()))
        b = m.generate(img.flip(0), tok, max_text_length=d['max_text_length'], top_k=1, graph=False)
        assert len(dec._sessions) == 1 and next(iter(dec._sessions.values())) is first      # re-bound, not rebuilt
        assert torch.equal(a[0], v['free_ids']) and torch.equal(b[0], v['free_ids'].flip(0))
        m.generate(img[:2], tok, max_text_length=d['max_text_length'], top_k=1, graph=False)  # another shape: a second one
        assert len(dec._sessions) == 2
        dec.transformer.ln_f.weight.mul_(1.5)                                                # new parameter state
        assert first.stale()
        m.generate(img, tok, max_text_length=d['max_text_length'], top_k=1, graph=False)
        assert all(not s.stale() for s in dec._sessions.values()) and first not in dec._sessions.values()


def test_sample_next_token_declines_what_the_kernel_does_not_take():
    """f32 logits (the parity configuration), CPU tensors and strided slices go to the framework ops: the fused sampler
    says so by returning None before touching the library."""
    from lavila_amd.narrator import sample_next_token
    assert sample_next_token(torch.randn(3, 50), 1, None, 1.0) is None                       # f32, CPU
    assert sample_next_token(torch.randn(3, 50).bfloat16(), 1, None, 1.0) is None            # bf16 but not on the device
    assert sample_next_token(torch.randn(3, 4, 50).bfloat16()[:, -1], None, 0.9, 0.7) is None


def test_narrator_modules_resolve_under_the_reference_import_paths():
    """`from lavila.models.narrator import VCLM_HF`, `lavila.models.gpt2_gated`, `lavila.models.coca` (models.py:15-17,
    narrator.py:26) give the MI355X-native classes."""
    import lavila.models.coca as coca
    import lavila.models.gpt2_gated as g
    import lavila.models.narrator as n
    import lavila_amd.gpt2_gated
    import lavila_amd.narrator
    assert n.VCLM_HF is lavila_amd.narrator.VCLM_HF
    assert g.GPT2LMHeadModel is lavila_amd.gpt2_gated.GPT2LMHeadModel and callable(g.augment_gpt2_config)
    assert coca.CrossAttention is lavila_amd.narrator.CrossAttention and coca.LayerNorm is lavila_amd.narrator.LayerNorm


def test_pretrained_gpt2_loader_covers_every_plain_parameter_or_raises():
    """models.py:919-923 writes every parameter of the plain GPT-2 into the gated decoder and fails on a name it cannot
    resolve. Both checkpoint spellings load (LM-head keys `transformer.h...` and the hub's raw `h...`); a missing, an
    unknown or a mis-shaped parameter raises instead of leaving random weights behind."""
    import pytest
    from lavila_amd.gpt2_gated import GPT2LMHeadModel, augment_gpt2_config, gpt2_config
    from lavila_amd.models import load_gpt2_weights
    cfg = augment_gpt2_config(gpt2_config('gpt2'), cross_attn_freq=1, gated_xattn=True)
    cfg.n_layer, cfg.n_embd, cfg.n_head, cfg.vocab_size, cfg.n_positions = 2, 64, 1, 97, 16
    dec = GPT2LMHeadModel(cfg)
    tags = ('crossattention', 'ln_cross_attn', 'alpha_cattn', 'alpha_dense')
    g = torch.Generator().manual_seed(0)
    plain = {n: torch.randn(p.shape, generator=g) for n, p in dec.named_parameters() if not any(t in n for t in tags)}
    plain['transformer.h.0.attn.bias'] = torch.ones(1, 1, 16, 16)          # HF buffers ride along in a state_dict
    plain['lm_head.weight'] = plain['transformer.wte.weight']
    for strip in (False, True):
        sd = {(k[len('transformer.'):] if strip and k.startswith('transformer.') else k): v for k, v in plain.items()}
        dec2 = GPT2LMHeadModel(cfg)
        loaded = load_gpt2_weights(dec2, sd)
        own = dict(dec2.named_parameters())
        assert all(torch.equal(own[k], plain[k]) for k in loaded) and len(loaded) == len(plain) - 2
    broken = dict(plain)
    del broken['transformer.h.1.mlp.c_fc.bias']
    with pytest.raises(RuntimeError, match='were not found'):
        load_gpt2_weights(GPT2LMHeadModel(cfg), broken)
    with pytest.raises(RuntimeError, match='no counterpart'):
        load_gpt2_weights(GPT2LMHeadModel(cfg), dict(plain, **{'transformer.h.7.ln_1.weight': torch.zeros(64)}))
    with pytest.raises(RuntimeError, match='expects'):
        load_gpt2_weights(GPT2LMHeadModel(cfg), dict(plain, **{'transformer.ln_f.weight': torch.zeros(65)}))


@pytest.mark.parametrize('variant', ['freq1_gated', 'freq2_plain'])
def test_beam_search_bookkeeping_matches_reference(variant, monkeypatch):
    """VCLM_HF.beam_sample / group_beam_search (narrator.py:149-366) against the reference's own runs
    (tests/golden/narrator_beam.pt: unmodified narrator.py + the transformers-4.27 BeamSearchScorer restated in
    oracle/beam_scorer.py): sequences equal, scores to rounding -- on the key/value cache (rows re-gathered with the beams)
    and on the reference's recompute schedule. The beam_sample cases use top_k = 2, where the draw of 2 * num_beams
    candidates returns all of them and the outcome does not depend on the random stream."""
    _emulate_device_primitives(monkeypatch)
    m, c, d, v = _model(variant)
    bx = load_golden('narrator_beam.pt')['variants'][variant]
    img = v['image_tokens']
    with torch.no_grad():
        for name, run in bx['runs'].items():
            tok = types.SimpleNamespace(bos_token_id=bx['bos'], eos_token_id=run['eos'], pad_token_id=bx['pad'])
            for cache in (True, False):
                for seed in (0, 123):
                    torch.manual_seed(seed)
                    seq, score = getattr(m, run['fn'])(img, tok, max_text_length=run['max_text_length'], kv_cache=cache,
                                                       graph=False, **run['kwargs'])
                    assert seq.shape == run['sequences'].shape and torch.equal(seq, run['sequences']), (name, cache, seed)
                    torch.testing.assert_close(score, run['sequence_scores'], atol=1e-4, rtol=1e-4)


def test_beam_scorer_equals_the_restated_transformers_class_on_random_candidates():
    """lavila_amd.beam_search.BeamScorer against oracle.beam_scorer.BeamSearchScorer (the 4.27 class restated) on random
    candidate streams with frequent eos: same beams after every process() call, same finalize()."""
    from lavila_amd.beam_search import BeamScorer
    from oracle.beam_scorer import BeamSearchScorer
    g = torch.Generator().manual_seed(3)
    for trial in range(30):
        entries, groups = int(torch.randint(1, 4, (1,), generator=g)), int(torch.randint(1, 3, (1,), generator=g))
        sub = int(torch.randint(1, 4, (1,), generator=g))
        beams = groups * sub
        if beams < 2:
            continue
        lp = [1.0, 0.5, 2.0][trial % 3]
        keep = int(torch.randint(1, beams + 1, (1,), generator=g))
        a = BeamSearchScorer(entries, beams, 'cpu', length_penalty=lp, num_beam_hyps_to_keep=keep, num_beam_groups=groups)
        b = BeamScorer(entries, beams, 'cpu', length_penalty=lp, keep=keep, num_beam_groups=groups)
        ids = torch.randint(5, 50, (entries * beams, 1), generator=g)
        scores = torch.zeros(entries * beams)
        eos, pad = 1, 0
        for step in range(6):
            cur = torch.zeros(entries * beams, dtype=torch.long)
            for gi in range(groups):
                rows = (torch.arange(entries)[:, None] * beams + torch.arange(gi * sub, gi * sub + sub)[None]).reshape(-1)
                gids = ids[rows]
                cs = -torch.rand(entries, 2 * sub, generator=g).cumsum(1) - step          # descending
                ct = torch.randint(1, 4, (entries, 2 * sub), generator=g)                # eos = 1 shows up often
                # never more than `sub` eos among the candidates (the scorer's own precondition)
                for e in range(entries):
                    seen = 0
                    for j in range(2 * sub):
                        if ct[e, j] == eos:
                            seen += 1
                            if seen > sub:
                                ct[e, j] = 2
                ci = torch.randint(0, sub, (entries, 2 * sub), generator=g)
                ra = a.process(gids, cs, ct, ci, pad_token_id=pad, eos_token_id=eos)
                rb = b.process(gids, cs, ct, ci, pad, eos)
                assert torch.equal(ra['next_beam_scores'], rb[0]) and torch.equal(ra['next_beam_tokens'], rb[1])
                assert torch.equal(ra['next_beam_indices'], rb[2])
                scores[rows] = rb[0]
                ids[rows] = gids[rb[2]]
                cur[rows] = rb[1]
            ids = torch.cat([ids, cur[:, None]], 1)
            assert bool(a.is_done) == b.is_done
        fa = a.finalize(ids, scores, None, None, max_length=7, pad_token_id=pad, eos_token_id=eos)
        fb = b.finalize(ids, scores, 7, pad, eos)
        assert torch.equal(fa['sequences'], fb[0]) and torch.equal(fa['sequence_scores'], fb[1])
