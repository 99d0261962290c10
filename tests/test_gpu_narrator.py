"""GPU (-m gpu): the narrator's decoder side (SURVEY.md 8f rank 4 / BASELINE configs[4]) through the C ABI --
the row kernels of lavila_amd/csrc/decode.hip against the oracle, and `VCLM_HF.forward` / `VCLM_HF.generate` on the
reference's own outputs (tests/golden/narrator_decoder.pt, generated from the unmodified narrator.py + gpt2_gated.py)."""
import contextlib
import io
import types

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import oracle as O
from lavila_amd.guards import forbid_library_gemm

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _tol(dt):
    return dict(atol=1e-5, rtol=1e-5) if dt == torch.float32 else dict(atol=2e-2, rtol=2e-2)


# ----------------------------------------------------------------------------------------------------------------------
# kernels
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('rows,L,D,vocab,positions', [(6, 3, 192, 331, 40), (64, 1, 768, 50257, 1024), (5, 5, 1600, 100, 8)])
def test_gpt2_embed(dt, rows, L, D, vocab, positions):
    from lavila_amd import _cabi as C
    g = torch.Generator().manual_seed(1)
    wte = torch.randn(vocab, D, generator=g).to(dt)
    wpe = torch.randn(positions, D, generator=g).to(dt)
    ids = torch.randint(0, vocab, (rows,), generator=g)
    ids_d, wte_d, wpe_d = ids.to(DEV), wte.to(DEV), wpe.to(DEV)
    for p0 in (None, 2):
        pos = None if p0 is None else torch.tensor([p0], dtype=torch.int32, device=DEV)
        out = torch.empty(rows, D, dtype=dt, device=DEV)
        C.check(C.lib().lvl_gpt2_embed(C.ptr(ids_d), C.ptr(wte_d), C.ptr(wpe_d), C.ptr(pos), C.ptr(out),
                                       rows, L, D, vocab, positions, C.dtype_code(out), C.stream_ptr()), 'embed')
        want = (wte[ids].float() + wpe[(p0 or 0) + torch.arange(rows) % L].float()).to(dt)
        assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('rows,D', [(3, 192), (64, 768), (7, 1600), (2, 4096), (5, 8)])
def test_gated_add_layernorm(dt, rows, D):
    from lavila_amd import _cabi as C
    g = torch.Generator().manual_seed(2)
    res = torch.randn(rows, D, generator=g).to(dt)
    y = (2 * torch.randn(rows, D, generator=g)).to(dt)
    gamma = 1 + 0.1 * torch.randn(D, generator=g)
    beta = 0.1 * torch.randn(D, generator=g)
    gate = torch.tensor([0.37])
    y_d, gate_d, gamma_d, beta_d = y.to(DEV), gate.to(DEV), gamma.to(DEV), beta.to(DEV)
    for use_y, use_gate, in_place in ((True, True, True), (True, False, False), (False, False, False)):
        r = res.to(DEV).clone()
        s = r if in_place else (torch.empty_like(r) if use_y else None)
        h = torch.empty_like(r)
        C.check(C.lib().lvl_gated_add_layernorm(C.ptr(r), C.ptr(y_d) if use_y else None,
                                                C.ptr(gate_d) if use_gate else None, C.ptr(gamma_d),
                                                C.ptr(beta_d), 1e-5, C.ptr(s), C.ptr(h), rows, D, C.dtype_code(r),
                                                C.stream_ptr()), 'gated_add_layernorm')
        want_s = res.float()
        if use_y:
            want_s = (want_s + (0.37 if use_gate else 1.0) * y.float()).to(dt).float()
        want_h = O.layer_norm(want_s, gamma, beta, 1e-5)
        if s is not None:
            torch.testing.assert_close(s.float().cpu(), want_s, **(dict(atol=1e-6, rtol=1e-6) if dt == torch.float32
                                                                   else dict(atol=1e-6, rtol=2 ** -7)))       # 1 bf16 ulp; exact cancellations differ by f32 rounding
        torch.testing.assert_close(h.float().cpu(), want_h, **_tol(dt))


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_decoder_activations(dt):
    from lavila_amd import _cabi as C
    x = (3 * torch.randn(40, 3072, generator=torch.Generator().manual_seed(3))).to(dt)
    for act, fn in ((C.ACT_GELU_NEW, O.gelu_new), (C.ACT_SQRELU, O.sq_relu)):
        u = x.to(DEV).clone()
        C.check(C.lib().lvl_act_inplace(C.ptr(u), u.numel(), act, C.dtype_code(u), C.stream_ptr()), 'act')
        torch.testing.assert_close(u.float().cpu(), fn(x.float()), **_tol(dt))
        want_torch = F.gelu(x.float(), approximate='tanh') if act == C.ACT_GELU_NEW else torch.relu(x.float()) ** 2
        torch.testing.assert_close(u.float().cpu(), want_torch, **_tol(dt))


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,H,steps,cap', [(2, 3, 9, 12), (3, 12, 70, 77), (1, 1, 1, 1), (2, 2, 40, 64)])
def test_decode_self_attention_steps(dt, B, H, steps, cap):
    """Every step appends its k | v row and attends to rows 0..pos: equals the LAST row of the causal attention over the
    prefix (gpt2_attention_core, gpt2_gated.py:206-238)."""
    from lavila_amd import _cabi as C
    D = H * 64
    g = torch.Generator().manual_seed(4)
    qkv_all = torch.randn(B, steps, 3 * D, generator=g).to(dt)
    cache = torch.full((B, cap, 2 * D), float('nan'), dtype=dt, device=DEV)       # unwritten rows must never be read
    pos = torch.zeros(1, dtype=torch.int32, device=DEV)
    f = qkv_all.float()
    want = O.gpt2_attention_core(f[..., :D], f[..., D:2 * D], f[..., 2 * D:], H, causal=True)
    for t in range(steps):
        qkv = qkv_all[:, t].contiguous().to(DEV)
        out = torch.empty(B, D, dtype=dt, device=DEV)
        C.check(C.lib().lvl_decode_self_attn(C.ptr(qkv), C.ptr(cache), C.ptr(pos), C.ptr(out), B, cap, H,
                                             C.dtype_code(qkv), C.stream_ptr()), 'decode_self_attn')
        pos.add_(1)
        torch.testing.assert_close(out.float().cpu(), want[:, t], **_tol(dt))
    assert torch.equal(cache[:, :steps].cpu(), qkv_all[..., D:])
    assert torch.isnan(cache[:, steps:].float()).all()


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('contexts,qrep,H,T', [(2, 1, 3, 24), (3, 12, 12, 256), (2, 5, 2, 37), (1, 3, 1, 1), (2, 20, 3, 200),
                                               (1, 76, 2, 256), (2, 16, 1, 16), (2, 4, 2, 300)])
def test_cross_attention_rows(dt, contexts, qrep, H, T):
    from lavila_amd import _cabi as C
    D = H * 64
    g = torch.Generator().manual_seed(5)
    q = torch.randn(contexts * qrep, D, generator=g).to(dt)
    kv = torch.randn(contexts, T, 2 * D, generator=g).to(dt)
    out = torch.empty(contexts * qrep, D, dtype=dt, device=DEV)
    q_d, kv_d = q.to(DEV), kv.to(DEV)
    C.check(C.lib().lvl_cross_attn_rows_fwd(C.ptr(q_d), C.ptr(kv_d), C.ptr(out), contexts * qrep, qrep, T, H,
                                            C.dtype_code(out), C.stream_ptr()), 'cross_attn_rows')
    kf = kv.float()
    want = O.gpt2_attention_core(q.float().reshape(contexts, qrep, D), kf[..., :D], kf[..., D:], H, causal=False)
    torch.testing.assert_close(out.float().cpu(), want.reshape(-1, D), **_tol(dt))


@pytest.mark.parametrize('act', [None, 'gelu_new', 'sqrelu'])
@pytest.mark.parametrize('M,N,K', [(1, 768, 768), (7, 2304, 768), (64, 768, 3072), (65, 192, 192), (200, 1600, 1600),
                                   (16, 48, 32), (64, 50432, 768), (130, 3072, 768), (150, 48, 64), (640, 768, 3072),
                                   (129, 2304, 768), (150, 64, 96), (333, 1600, 1600), (131, 128, 64), (150, 96, 128),
                                   (257, 768, 768)])
def test_skinny_gemm(M, N, K, act):
    """lvl_linear_skinny against the f32 product of the same bf16 operands: one bf16 rounding of the result (2^-9
    relative) plus f32 accumulation-order noise; every row / column / k-step remainder of the 16- and 32-row blocks, 16-
    and 32-column strips, the 8-way contraction split and its 6-step rounds, and of the LDS-staged 64 x 64 / 64 x 128
    tiles (beyond 128 rows, or 50432 columns) is hit by the shapes above."""
    from lavila_amd import _cabi as C
    if act is not None and N > 4096:
        pytest.skip('activation variants are covered on the smaller shapes')
    g = torch.Generator().manual_seed(M * 131 + N)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g)
    x_d, w_d, b_d = x.to(DEV), w.to(DEV), b.to(DEV)
    code = {None: -1, 'gelu_new': C.ACT_GELU_NEW, 'sqrelu': C.ACT_SQRELU}[act]
    for bias in (b_d, None):
        y = torch.full((M + 1, N), float('nan'), dtype=torch.bfloat16, device=DEV)       # a guard row below the result
        C.check(C.lib().lvl_linear_skinny(C.ptr(x_d), C.ptr(w_d), C.ptr(bias), C.ptr(y), M, N, K, code, C.stream_ptr()),
                'lvl_linear_skinny')
        want = x_d.float() @ w_d.float().t() + (0 if bias is None else bias)
        want = {None: lambda t: t, 'gelu_new': O.gelu_new, 'sqrelu': O.sq_relu}[act](want.cpu())
        torch.testing.assert_close(y[:M].float().cpu(), want, atol=1e-2, rtol=1e-2)
        assert torch.isnan(y[M].float()).all()
    assert C.lib().lvl_linear_skinny(C.ptr(x_d), C.ptr(w_d), None, C.ptr(y), M, N - 8, K, -1, C.stream_ptr()) == -38
    assert C.lib().lvl_linear_skinny(C.ptr(x_d), C.ptr(w_d), None, C.ptr(y), M, N, K - 16, -1, C.stream_ptr()) == -38




@pytest.mark.parametrize('M,N,K', [(64, 768, 768), (64, 3072, 768), (7, 2304, 768), (128, 192, 192), (33, 1600, 1600),
                                   (64, 4800, 1600), (1, 48, 32), (100, 256, 1024)])
def test_skinny_gemm_with_layernorm_prologue(M, N, K):
    """lvl_linear_skinny_ln == lvl_gated_add_layernorm followed by lvl_linear_skinny: the new residual to the bit (same
    fma, same rounding), the product within one bf16 rounding of the normalised operand (the row statistics are summed
    in another order); and against the f32 formula."""
    from lavila_amd import _cabi as C
    g = torch.Generator().manual_seed(M + N + K)
    res = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    y = (2 * torch.randn(M, K, generator=g)).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    gamma = (1 + 0.1 * torch.randn(K, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(K, generator=g)).to(DEV)
    gate = torch.tensor([0.37], device=DEV)
    for use_y, use_gate, act in ((True, True, -1), (True, False, C.ACT_GELU_NEW), (False, False, C.ACT_SQRELU)):
        out = torch.full((M + 1, N), float('nan'), dtype=torch.bfloat16, device=DEV)
        new_res = torch.full((M + 1, K), float('nan'), dtype=torch.bfloat16, device=DEV)
        C.check(C.lib().lvl_linear_skinny_ln(C.ptr(res), C.ptr(y) if use_y else None, C.ptr(gate) if use_gate else None,
                                             C.ptr(gamma), C.ptr(beta), 1e-5, C.ptr(new_res) if use_y else None, C.ptr(w),
                                             C.ptr(b), C.ptr(out), M, N, K, act, C.stream_ptr()), 'lvl_linear_skinny_ln')
        s_ref, h_ref = res.clone(), torch.empty_like(res)
        C.check(C.lib().lvl_gated_add_layernorm(C.ptr(s_ref), C.ptr(y) if use_y else None, C.ptr(gate) if use_gate else None,
                                                C.ptr(gamma), C.ptr(beta), 1e-5, C.ptr(s_ref) if use_y else None,
                                                C.ptr(h_ref), M, K, C.LVL_BF16, C.stream_ptr()), 'gated_add_layernorm')
        o_ref = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        C.check(C.lib().lvl_linear_skinny(C.ptr(h_ref), C.ptr(w), C.ptr(b), C.ptr(o_ref), M, N, K, act, C.stream_ptr()), 'skinny')
        if use_y:
            assert torch.equal(new_res[:M], s_ref)
            assert torch.isnan(new_res[M].float()).all()
        torch.testing.assert_close(out[:M].float(), o_ref.float(), atol=3e-2, rtol=3e-2)
        assert (out[:M].float() - o_ref.float()).abs().mean().item() < 2e-3
        assert torch.isnan(out[M].float()).all()
        hf = O.layer_norm(s_ref.float().cpu(), gamma.cpu(), beta.cpu(), 1e-5)
        want = hf @ w.float().cpu().t() + b.cpu()
        want = {-1: lambda t: t, C.ACT_GELU_NEW: O.gelu_new, C.ACT_SQRELU: O.sq_relu}[act](want)
        torch.testing.assert_close(out[:M].float().cpu(), want, atol=4e-2, rtol=4e-2)
    other = torch.empty_like(res)
    assert C.lib().lvl_linear_skinny_ln(C.ptr(res), C.ptr(y), None, C.ptr(gamma), C.ptr(beta), 1e-5, C.ptr(res), C.ptr(w),
                                        C.ptr(b), C.ptr(out), M, N, K, -1, C.stream_ptr()) == -22       # in place: refused
    assert C.lib().lvl_linear_skinny_ln(C.ptr(res), C.ptr(y), None, C.ptr(gamma), C.ptr(beta), 1e-5, C.ptr(other), C.ptr(w),
                                        C.ptr(b), C.ptr(out), M, N, 2048, -1, C.stream_ptr()) == -38     # K too long


# ----------------------------------------------------------------------------------------------------------------------
# the sampling kernel
# ----------------------------------------------------------------------------------------------------------------------
def _padded_logits(rows, vocab, seed, scale=3.0):
    g = torch.Generator().manual_seed(seed)
    pad = -vocab % 256
    full = torch.full((rows, vocab + pad), 1e4)                  # padded columns hold junk the kernel must ignore
    full[:, :vocab] = scale * torch.randn(rows, vocab, generator=g)
    return full.bfloat16().to(DEV)[:, :vocab]


@pytest.mark.parametrize('vocab', [50257, 331, 8])
def test_sampler_perplexity_terms_and_greedy(vocab):
    """nll = entropy of the unwarped softmax, or the cross entropy against a target with pad ignored (narrator.py:
    124-131); top_k = 1 draws the argmax whatever the uniform."""
    from lavila_amd.narrator import sample_next_token
    rows = 37
    logits = _padded_logits(rows, vocab, 1)
    g = torch.Generator().manual_seed(2)
    best = torch.randint(0, vocab, (rows,), generator=g).to(DEV)
    logits[torch.arange(rows), best] = 20.0                      # a unique maximum per row
    f = logits.float()
    nxt, nll, cnt = sample_next_token(logits, 1, None, 1.0)
    assert torch.equal(nxt[:, 0], best) and nxt.shape == (rows, 1) and nxt.dtype == torch.int64
    torch.testing.assert_close(nll, torch.special.entr(F.softmax(f, dim=1)).sum(1), atol=2e-4, rtol=2e-4)
    assert torch.equal(cnt, torch.ones(rows, device=DEV))
    target = torch.randint(0, vocab, (rows,), generator=g).to(DEV)
    target[::5] = 0                                              # pad id 0
    nxt, nll, cnt = sample_next_token(logits, 1, 0.9, 0.7, target=target, pad_id=0)
    assert torch.equal(nxt[:, 0], best)
    torch.testing.assert_close(nll, F.cross_entropy(f, target, ignore_index=0, reduction='none'), atol=2e-4, rtol=2e-4)
    assert torch.equal(cnt, target.ne(0).float())


def _expected_draw(f, lo, rdrop, vstar, inv_t, u):
    """The kernel's rule restated in float64: kept = value >= lo, minus the first `rdrop` (index order) entries equal to
    the boundary value; inverse CDF in index order."""
    lo = float('-inf') if lo != lo else lo                      # no threshold: the kernel reports the key-0 sentinel (NaN)
    keep = f >= lo
    if rdrop > 0:
        ties = (f == vstar).nonzero()[:, 0]
        keep[ties[:int(rdrop)]] = False
    mass = torch.where(keep, torch.exp((f.double() - f.max().double()) * inv_t), torch.zeros_like(f, dtype=torch.float64))
    cum = mass.cumsum(0)
    return keep, cum, u * cum[-1]


@pytest.mark.parametrize('vocab,top_k,top_p,temp', [(50257, None, 0.95, 0.7), (50257, 50, None, 1.0), (50257, 40, 0.9, 0.8),
                                                    (331, None, 0.5, 1.0), (331, 5, 0.3, 2.0), (1000, None, 0.999, 1.0),
                                                    (64, 64, 1.0, 1.0), (50257, None, None, 1.3)])
def test_sampler_kept_set_and_draw(vocab, top_k, top_p, temp):
    """The kept set equals transformers' warpers' (VCLM_HF._warp restates them; entries tied with the top-p boundary
    value may differ in WHICH of them go, not in how many), and the token is the inverse CDF of the kept entries at the
    given uniform."""
    from lavila_amd.narrator import VCLM_HF, sample_next_token
    rows = 12
    logits = _padded_logits(rows, vocab, 7 + vocab, scale=2.5)
    u = torch.rand(rows, generator=torch.Generator().manual_seed(3)).to(DEV)
    nxt, nll, cnt, dbg = sample_next_token(logits, top_k, top_p, temp, uniform=u, debug=True)
    f = logits.float().cpu()
    ref = VCLM_HF._warp(logits.float(), top_k, top_p, temp).cpu() > float('-inf')
    for r in range(rows):
        lo, rdrop, zk, vstar = dbg[r, :4].tolist()
        keep, cum, want = _expected_draw(f[r], lo, rdrop, vstar, 1.0 / temp, u[r].item())
        assert abs(int(keep.sum()) - int(ref[r].sum())) <= 2, (r, int(keep.sum()), int(ref[r].sum()))
        differ = (keep != ref[r]).nonzero()[:, 0]
        assert all(f[r, i] == vstar or f[r, i] == f[r][ref[r]].min() for i in differ), (r, differ, vstar)
        assert abs(zk - cum[-1].item()) <= 1e-4 * cum[-1].item()
        t = int(nxt[r, 0])
        assert keep[t], (r, t)
        eps = 2e-5 * cum[-1].item()
        assert (cum[t - 1].item() if t > 0 else 0.0) - eps <= want <= cum[t].item() + eps, (r, t)


def test_sampler_draws_follow_the_distribution():
    """20000 draws of one row at nucleus 0.9 / temperature 0.8: frequencies match the renormalised kept probabilities."""
    from lavila_amd.narrator import VCLM_HF, sample_next_token
    vocab, n = 200, 20000
    row = _padded_logits(1, vocab, 11, scale=1.5)
    logits = row.expand(n, vocab)
    pad = -vocab % 256
    logits = torch.cat([logits, torch.zeros(n, pad, dtype=torch.bfloat16, device=DEV)], 1)[:, :vocab]
    torch.manual_seed(0)
    nxt, _, _ = sample_next_token(logits, None, 0.9, 0.8)
    want = F.softmax(VCLM_HF._warp(row.float(), None, 0.9, 0.8), dim=-1)[0]
    freq = torch.bincount(nxt[:, 0], minlength=vocab).float() / n
    assert freq[want == 0].sum().item() <= 2.0 / n * 50          # nothing (beyond boundary ties) outside the nucleus
    assert (freq - want).abs().max().item() < 4 * (want.max().item() / n) ** 0.5 + 2e-3


def test_generate_with_fused_sampler_matches_framework_ops_greedy(monkeypatch):
    """bf16 decode, top_k=1: the fused sampler and the framework ops pick the same tokens and perplexities."""
    m, c, d, w = _mid_model('bf16')
    m = m.bfloat16()
    tok = types.SimpleNamespace(bos_token_id=d['vocab'] - 1, eos_token_id=7, pad_token_id=0)
    g = torch.Generator().manual_seed(12)
    img = torch.randn(3, c['queries'], c['text_width'], generator=g).to(DEV).bfloat16()
    with torch.no_grad():
        a = m.generate(img, tok, max_text_length=12, top_k=1, num_return_sequences=2)
        monkeypatch.setenv('LAVILA_NARRATOR_SAMPLER', 'torch')
        b = m.generate(img, tok, max_text_length=12, top_k=1, num_return_sequences=2)
    assert torch.equal(a[0], b[0])
    torch.testing.assert_close(a[1], b[1], atol=0, rtol=2e-3)


# ----------------------------------------------------------------------------------------------------------------------
# the model on the reference's outputs
# ----------------------------------------------------------------------------------------------------------------------
def _build(c, d, var, width=None, heads=None):
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeTransformer
    from lavila_amd.gpt2_gated import GPT2LMHeadModel, augment_gpt2_config, gpt2_config
    from lavila_amd.narrator import VCLM_HF
    width, heads = width or c['text_width'], heads or c['pool_heads']
    with contextlib.redirect_stdout(io.StringIO()):
        vis = SpaceTimeTransformer(img_size=c['img'], patch_size=c['patch'], embed_dim=c['dim'], depth=c['depth'],
                                   num_heads=c['heads'], num_frames=c['frames'], time_init='zeros',
                                   attention_style='frozen-in-time', ln_pre=True, act_layer=QuickGELU)
    vis.head = vis.pre_logits = vis.fc = torch.nn.Identity()
    base = gpt2_config('gpt2', vocab_size=d['vocab'], n_positions=d['positions'], n_embd=width, n_layer=d['layers'],
                       n_head=heads)
    dec = GPT2LMHeadModel(augment_gpt2_config(base, **var))
    return VCLM_HF(vision_width=c['dim'], vision_model=vis, text_width=width, text_decoder=dec,
                   num_img_queries=c['queries'], dim_head=64, heads=heads)


def _golden_model(variant):
    fx = load_golden('narrator_decoder.pt')
    c, d, v = fx['config'], fx['decoder'], fx['variants'][variant]
    m = _build(c, d, v['variant'])
    w = O.narrator_weights(v['shapes'], seed=v['weight_seed'])
    own = m.state_dict()
    assert set(v['shapes']) | set(v['kept_buffers']) | {'text_decoder.lm_head.weight'} == set(own)
    for k in v['kept_buffers']:                     # causal-mask buffers / beta zeros keep their constructed values
        w[k] = own[k]
    m.load_state_dict(w, strict=True)
    assert m.text_decoder.lm_head.weight is m.text_decoder.transformer.wte.weight
    video, _ = O.synthetic_batch(c['batch'], c['frames'], c['img'], seed=v['input_seed'])
    tok = types.SimpleNamespace(bos_token_id=v['bos'], eos_token_id=v['eos'], pad_token_id=v['pad'])
    return m.to(DEV).eval(), c, d, v, video.to(DEV), tok


@pytest.mark.parametrize('variant', ['freq1_gated', 'freq2_plain'])
def test_narrator_forward_matches_reference_f32(variant):
    """VCLM_HF.forward (narrator.py:89-104): teacher-forced logits [B, V, L-1] and labels, float32 within 1e-3 of the
    reference's CPU output (logits of a few units)."""
    m, c, d, v, video, tok = _golden_model(variant)
    with torch.no_grad(), forbid_library_gemm():        # round 5: the float32 decoder runs on the own f32-class kernels
        out = m(video, v['text'].to(DEV))
    assert out['text_tokens_logits'].dtype == torch.float32
    assert torch.equal(out['labels'].cpu(), v['labels'])
    torch.testing.assert_close(out['text_tokens_logits'].cpu(), v['logits'], atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('variant', ['freq1_gated', 'freq2_plain'])
def test_narrator_generate_matches_reference_f32(variant, graph):
    """VCLM_HF.generate(top_k=1) (narrator.py:106-147) on the reference's own ids / perplexities: free running, with a
    live eos, early stopping, target scoring with and without teacher forcing, num_return_sequences -- decoded against the
    key/value cache (eagerly and as hipGraph replays), and once more on the reference's recompute schedule."""
    m, c, d, v, video, tok = _golden_model(variant)
    no_eos = types.SimpleNamespace(bos_token_id=v['bos'], eos_token_id=-1, pad_token_id=v['pad'])
    text = v['text'].to(DEV)
    with torch.no_grad(), forbid_library_gemm():
        img = m.encode_image(video)
        torch.testing.assert_close(img.cpu(), v['image_tokens'], atol=1e-3, rtol=1e-3)
        runs = [
            ('free', dict(tokenizer=no_eos, max_text_length=d['max_text_length'])),
            ('eos', dict(tokenizer=tok, max_text_length=d['max_text_length'])),
            ('tf', dict(tokenizer=tok, target=text, max_text_length=d['text_len'], teacher_forcing=True)),
            ('tgt', dict(tokenizer=tok, target=text, max_text_length=d['text_len'])),
            ('rep', dict(tokenizer=tok, max_text_length=8, num_return_sequences=2)),
        ]
        for name, kw in runs:
            for cache in ((True,) if graph else (True, False)):
                ids, ppl = m.generate(img, top_k=1, kv_cache=cache, graph=graph, **kw)
                assert torch.equal(ids.cpu(), v[name + '_ids']), (name, cache)
                torch.testing.assert_close(ppl.cpu(), v[name + '_ppl'], atol=0, rtol=5e-3, msg=lambda s: f'{name} {cache}: {s}')
        ids, ppl = m.generate(img[:1], tok, max_text_length=d['max_text_length'], top_k=1, early_stopping=True, graph=graph)
        assert torch.equal(ids.cpu(), v['stop_ids'])
        torch.testing.assert_close(ppl.cpu(), v['stop_ppl'], atol=0, rtol=5e-3)


@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('variant', ['freq1_gated', 'freq2_plain'])
def test_narrator_beam_search_matches_reference_f32(variant, graph):
    """VCLM_HF.beam_sample / group_beam_search (narrator.py:149-366) on the HIP path against the reference's own runs
    (tests/golden/narrator_beam.pt): sequences equal, scores within f32 round-off -- on the key/value cache whose rows are
    re-gathered with the beams after every step (eagerly and under hipGraph replay) and on the recompute schedule."""
    m, c, d, v, video, tok = _golden_model(variant)
    bx = load_golden('narrator_beam.pt')['variants'][variant]
    with torch.no_grad(), forbid_library_gemm():
        img = m.encode_image(video)
        for name, run in bx['runs'].items():
            tk = types.SimpleNamespace(bos_token_id=bx['bos'], eos_token_id=run['eos'], pad_token_id=bx['pad'])
            for cache in ((True,) if graph else (True, False)):
                torch.manual_seed(7)
                seq, score = getattr(m, run['fn'])(img, tk, max_text_length=run['max_text_length'], kv_cache=cache,
                                                   graph=graph, **run['kwargs'])
                assert torch.equal(seq.cpu(), run['sequences']), (name, cache, seq.cpu(), run['sequences'])
                torch.testing.assert_close(score.cpu(), run['sequence_scores'], atol=2e-3, rtol=2e-3,
                                           msg=lambda s: f'{name} {cache}: {s}')


def test_narrator_beam_search_bf16_runs_and_ranks_its_beams():
    """bf16 decoder (the captioning recipe's precision): beam searches run on the cached session, return the requested
    number of sequences, best first, with finite scores; a sampled beam search with a wide top_k stays inside the
    vocabulary and differs between seeds only through the draw."""
    m, c, d, w = _mid_model('bf16')
    m = m.bfloat16()
    g = torch.Generator().manual_seed(4)
    enc = torch.randn(3, c['queries'], c['text_width'], generator=g).to(DEV).bfloat16()
    tk = types.SimpleNamespace(bos_token_id=d['vocab'] - 1, eos_token_id=5, pad_token_id=0)
    with torch.no_grad():
        seq, score = m.group_beam_search(enc, tk, max_text_length=10, num_beams=6, num_beam_groups=3, num_return_sequences=3)
        assert seq.shape[0] == 9 and seq.shape[1] <= 10 and torch.isfinite(score).all()
        sc = score.view(3, 3)
        assert bool((sc[:, :-1] >= sc[:, 1:]).all())                    # best first per clip
        seq2, score2 = m.beam_sample(enc, tk, max_text_length=10, num_beams=3, top_k=40, temperature=0.9)
        assert seq2.shape[0] == 3 and int(seq2.max()) < d['vocab'] and torch.isfinite(score2).all()


def test_narrator_sampling_warpers_and_shapes():
    """top-k / top-p / temperature restate transformers' warpers (narrator.py:368-389); multinomial sampling stays inside
    the kept set and the outputs have the reference's shapes."""
    from lavila_amd.narrator import VCLM_HF
    g = torch.Generator().manual_seed(6)
    logits = torch.randn(4, 50, generator=g).to(DEV)
    w = VCLM_HF._warp(logits, top_k=5, top_p=None, temperature=0.7)
    assert ((w > float('-inf')).sum(-1) == 5).all()
    kept = torch.topk(logits, 5).indices
    assert torch.equal(torch.sort((w > float('-inf')).nonzero()[:, 1].reshape(4, 5)).values, torch.sort(kept).values)
    torch.testing.assert_close(w.gather(1, kept), logits.gather(1, kept) / 0.7)
    w = VCLM_HF._warp(logits, top_k=None, top_p=0.6, temperature=1.0)
    p = logits.softmax(-1)
    for r in range(4):
        keep = w[r] > float('-inf')
        order = torch.argsort(p[r], descending=True)
        k = int(keep.sum())
        assert torch.equal(torch.sort(order[:k]).values, keep.nonzero()[:, 0])       # a prefix of the sorted probabilities
        assert p[r][order[:k]].sum() >= 0.6 - 1e-6 and (k == 1 or p[r][order[:k - 1]].sum() < 0.6 + 1e-6)
    m, c, d, v, video, tok = _golden_model('freq1_gated')
    with torch.no_grad():
        img = m.encode_image(video)
        torch.manual_seed(0)
        ids, ppl = m.generate(img, tok, max_text_length=10, top_k=3, top_p=0.9, temperature=0.8, num_return_sequences=3)
    assert ids.shape == (3 * c['batch'], 10) and ppl.shape == (3 * c['batch'],)
    assert (ids[:, 0] == v['bos']).all() and ids.max() < d['vocab'] and torch.isfinite(ppl).all()


def _mid_model(dtype_mode, width=256, heads=4, layers=2, vocab=331, freq=1):
    """A decoder whose widths tile the own GEMM (out % 256, in % 64) with procedural weights; the oracle is its checker."""
    fx = load_golden('narrator_decoder.pt')
    c = dict(fx['config'])
    c.update(text_width=width, pool_heads=heads)
    d = dict(fx['decoder'], layers=layers, vocab=vocab)
    m = _build(c, d, dict(cross_attn_freq=freq, gated_xattn=True), width, heads)
    sd = m.state_dict()
    keep = [k for k in sd if k.endswith('.attn.bias') or k.endswith('.crossattention.bias') or k.endswith('masked_bias')
            or k.endswith('.beta')]
    shapes = {k: tuple(t.shape) for k, t in sd.items() if k not in keep and k != 'text_decoder.lm_head.weight'}
    w = O.narrator_weights(shapes, seed=41)
    m.load_state_dict({**w, **{k: sd[k] for k in keep}}, strict=True)
    return m.to(DEV).eval(), c, d, w


@pytest.mark.parametrize('mode', ['bf16', 'half', 'autocast'])
def test_decoder_low_precision_on_own_gemms(mode, monkeypatch):
    """bf16 parameters, the --use-half recipe (fp16 parameters, docs/PRETRAIN.md:85-91) and f32 masters under autocast:
    every Conv1D of the decoder runs on lvl_linear_skinny / lvl_linear_skinny_ln (few rows) or lvl_linear_tn (many rows;
    forced here by setting the row thresholds to 0) -- never the library GEMM; teacher-forced logits and cached decoding agree with the f32 oracle
    within the bf16 bound on both kernels, cached == recomputed."""
    from lavila_amd import gpt2_gated as G
    from lavila_amd import ops
    m, c, d, w = _mid_model(mode)
    H = c['pool_heads']
    g = torch.Generator().manual_seed(9)
    B, L, NQ = 3, 11, c['queries']
    ids = torch.randint(1, d['vocab'], (B, L), generator=g)
    enc = torch.randn(B, NQ, c['text_width'], generator=g)
    want, _ = O.gpt2_lm_logits(ids, enc, w, H, prefix='text_decoder.')
    from lavila_amd import _cabi as C
    calls = {'tn': 0, 'skinny': 0, 'fused': 0}
    real_tn, real_sk, real_fused = ops.linear_tn_raw, G._Pack._skinny, C.lib().lvl_linear_skinny_ln
    monkeypatch.setattr(ops, 'linear_tn_raw', lambda *a, **k: (calls.__setitem__('tn', calls['tn'] + 1), real_tn(*a, **k))[1])
    monkeypatch.setattr(G._Pack, '_skinny', lambda *a, **k: (calls.__setitem__('skinny', calls['skinny'] + 1), real_sk(*a, **k))[1])
    monkeypatch.setattr(C.lib(), 'lvl_linear_skinny_ln',
                        lambda *a: (calls.__setitem__('fused', calls['fused'] + 1), real_fused(*a))[1])
    monkeypatch.setattr(F, 'linear', lambda *a, **k: (_ for _ in ()).throw(AssertionError('library GEMM in the decoder')))
    dec = m.text_decoder
    ctx = contextlib.nullcontext()
    if mode == 'bf16':
        dec = dec.bfloat16()
        enc_dev = enc.to(DEV).bfloat16()
    elif mode == 'half':
        dec = dec.half()
        enc_dev = enc.to(DEV).half()
    else:
        enc_dev = enc.to(DEV)
        ctx = torch.autocast('cuda', dtype=torch.bfloat16)
    n_gemms = d['layers'] * 9 + 1                         # 4 + 4 Conv1Ds and the image k|v projection per block, lm_head
    scale = want.abs().max().item()
    with torch.no_grad(), ctx:
        full = {}
        for kernel, rows, fused_rows in (('skinny', G.SKINNY_MAX_ROWS, G.FUSED_LN_MAX_ROWS), ('tn', 0, 0)):
            monkeypatch.setattr(G, 'SKINNY_MAX_ROWS', rows)
            monkeypatch.setattr(G, 'FUSED_LN_MAX_ROWS', fused_rows)
            calls.update(tn=0, skinny=0, fused=0)
            got = dec(ids.to(DEV), encoder_hidden_states=enc_dev).logits
            assert got.dtype == {'bf16': torch.bfloat16, 'half': torch.float16, 'autocast': torch.bfloat16}[mode]
            if kernel == 'skinny':       # per block 4 LN-fed Conv1Ds fused with their LayerNorm, 4 plain + the image k|v
                assert calls == {'skinny': d['layers'] * 5 + 1, 'fused': d['layers'] * 4, 'tn': 0}, calls
            else:
                assert calls == {'tn': n_gemms, 'skinny': 0, 'fused': 0}, calls
            assert (got.float().cpu() - want).abs().max().item() < 0.04 * scale
            full[kernel] = got
            for graph in (False, True):
                sess = dec.decode_session(enc_dev, L, graph=graph)
                for t in range(L):
                    step = sess.step(ids[:, t].to(DEV)).float().cpu()
                    assert (step - want[:, t]).abs().max().item() < 0.04 * scale, (kernel, graph, t)
                    assert (step - got[:, t].float().cpu()).abs().max().item() < 0.04 * scale
                assert not sess.stale()
                with pytest.raises(RuntimeError):
                    sess.step(ids[:, 0].to(DEV))             # the cache is full
            dec._sessions.clear()                            # the other kernel needs its own captured graph


def test_decode_session_graph_equals_eager_bitwise_and_tracks_weights(monkeypatch):
    """The captured step replays the eager step's kernels: identical bits (with the LayerNorm folding that only the
    eagerly launched step uses switched off; with it the logits agree within bf16 rounding); a parameter write makes the
    session stale and the next forward rebuilds the packed weights."""
    from lavila_amd import gpt2_gated as G
    m, c, d, w = _mid_model('bf16')
    dec = m.text_decoder.bfloat16()
    g = torch.Generator().manual_seed(10)
    ids = torch.randint(1, d['vocab'], (4, 7), generator=g).to(DEV)
    enc = torch.randn(2, c['queries'], c['text_width'], generator=g).to(DEV).bfloat16()
    with torch.no_grad():
        folded = dec.decode_session(enc, 7, seqs_per_context=2, graph=False)
        ref = [folded.step(ids[:, t]).float().clone() for t in range(7)]
        dec._sessions.clear()
        monkeypatch.setattr(G, 'FUSED_LN_MAX_ROWS', 0)
        a = dec.decode_session(enc, 7, seqs_per_context=2, graph=False)
        b = dec.decode_session(enc, 7, seqs_per_context=2, graph=True)
        for t in range(7):
            la, lb = a.step(ids[:, t]), b.step(ids[:, t])
            assert torch.equal(la, lb)
            assert (la.float() - ref[t]).abs().max().item() < 0.03 * ref[t].abs().max().item()
        full = dec(ids, encoder_hidden_states=enc.repeat_interleave(2, dim=0)).logits
        before = full.clone()
        dec.transformer.ln_f.weight.mul_(2.0)           # in place under no_grad: bumps the version the pack is keyed on
        assert a.stale() and b.stale()
        after = dec(ids, encoder_hidden_states=enc.repeat_interleave(2, dim=0)).logits
        assert not torch.equal(before, after)


def test_gpt2_xl_widths_decode(monkeypatch):
    """GPT-2 XL's layout (width 1600 = 25 heads, cross-attention in every 2nd block: models.py:981-1005) at 3 blocks:
    widths lvl_linear_tn does not tile. Decoding runs on the strip kernels (no library GEMM), the teacher-forced pass on
    few rows too; both agree with the f32 oracle and with each other."""
    m, c, d, w = _mid_model('bf16', width=1600, heads=25, layers=3, vocab=331, freq=2)
    dec = m.text_decoder.bfloat16()
    H = 25
    g = torch.Generator().manual_seed(13)
    B, L = 2, 9
    ids = torch.randint(1, d['vocab'], (B, L), generator=g)
    enc = torch.randn(B, c['queries'], 1600, generator=g)
    want, _ = O.gpt2_lm_logits(ids, enc, w, H, prefix='text_decoder.')
    monkeypatch.setattr(F, 'linear', lambda *a, **k: (_ for _ in ()).throw(AssertionError('library GEMM in the decoder')))
    scale = want.abs().max().item()
    with torch.no_grad():
        enc_dev = enc.to(DEV).bfloat16()
        got = dec(ids.to(DEV), encoder_hidden_states=enc_dev).logits.float().cpu()
        assert (got - want).abs().max().item() < 0.04 * scale
        for graph in (False, True):
            sess = dec.decode_session(enc_dev, L, graph=graph)
            for t in range(L):
                step = sess.step(ids[:, t].to(DEV)).float().cpu()
                assert (step - want[:, t]).abs().max().item() < 0.04 * scale, (graph, t)


def test_decoder_is_loud_about_what_it_does_not_do():
    m, c, d, w = _mid_model('f32')
    dec = m.text_decoder
    ids = torch.zeros(1, 4, dtype=torch.long, device=DEV)
    with pytest.raises(NotImplementedError):
        dec(ids)                                                     # gradients enabled on trainable parameters
    with torch.no_grad():
        for kw in (dict(attention_mask=torch.ones(1, 4, device=DEV)), dict(use_cache=True),
                   dict(past_key_values=((None, None),))):
            with pytest.raises(NotImplementedError):
                dec(ids, **kw)
        with pytest.raises(ValueError):
            dec(torch.zeros(1, d['positions'] + 1, dtype=torch.long, device=DEV))
        with pytest.raises(IndexError):
            dec(torch.full((1, 3), d['vocab'], dtype=torch.long, device=DEV))
        plain = dec(ids).logits                                      # no image tokens: plain GPT-2 (gpt2_gated.py:432)
        want, _ = O.gpt2_lm_logits(ids.cpu(), None, w, c['pool_heads'], prefix='text_decoder.')
        torch.testing.assert_close(plain.cpu(), want, atol=2e-3, rtol=1e-3)
        from lavila_amd._cabi import HipExtensionError
        with pytest.raises(HipExtensionError):
            m.text_decoder.cpu()(ids.cpu())                          # no CPU fallback
