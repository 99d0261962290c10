"""GPU (-m gpu): every C-ABI kernel against the CPU oracle on the same seeded inputs.

Tolerances: float32 path = f32 round-off of a different summation order (<= 2e-4 abs on O(1) values; the
north_star bar is 1e-3); bfloat16 path = inputs rounded to bf16 first, oracle evaluated in f32 on the rounded
inputs, so the allowance covers output rounding (2^-9 relative) and bf16-rounded intermediates."""
import pytest
import torch

from conftest import load_golden
from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda'
TOL = {torch.float32: dict(atol=2e-4, rtol=2e-4), torch.bfloat16: dict(atol=3e-2, rtol=3e-2)}


def _r(x, dt):
    """round to dt and come back to f32 on CPU (what the oracle sees)"""
    return x.to(dt).float()


def _close(got, want, dt, scale=1.0, msg=''):
    t = TOL[dt]
    torch.testing.assert_close(got.float().cpu(), want, atol=t['atol'] * scale, rtol=t['rtol'], msg=lambda m: f'{msg}: {m}')


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('rows,cols,eps', [(37, 128, 1e-6), (4, 768, 1e-5), (1031, 768, 1e-6), (5, 1024, 1e-6),
                                           (3, 512, 1e-5), (2, 3072, 1e-5), (0, 768, 1e-5)])
def test_layernorm_fwd_bwd(dt, rows, cols, eps):
    from lavila_amd import ops
    g = torch.Generator().manual_seed(rows * 7 + cols)
    x = _r(torch.randn(rows, cols, generator=g) * 2 + 0.5, dt)
    w = 1 + 0.2 * torch.randn(cols, generator=g)
    b = 0.1 * torch.randn(cols, generator=g)
    dy = _r(torch.randn(rows, cols, generator=g), dt)
    xo, wo, bo = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = O.layer_norm(xo, wo, bo, eps)
    yo.backward(dy)
    xg = x.to(DEV, dt).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.layer_norm(xg, wg, bg, eps)
    assert y.dtype == dt and y.shape == x.shape
    if rows == 0:
        return
    y.backward(dy.to(DEV, dt))
    _close(y, yo.detach(), dt, 4, 'y')
    _close(xg.grad, xo.grad, dt, 4, 'dx')
    _close(wg.grad, wo.grad, dt, 0.05 * rows + 1, 'dgamma')
    _close(bg.grad, bo.grad, dt, 0.05 * rows + 1, 'dbeta')


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('keep_sum,with_bias', [(True, True), (False, True), (True, False), (False, False)])
def test_add_layernorm_fused(dt, keep_sum, with_bias):
    from lavila_amd import ops
    rows, cols, eps = 77, 768, 1e-6
    g = torch.Generator().manual_seed(11)
    res, y = _r(torch.randn(rows, cols, generator=g), dt), _r(torch.randn(rows, cols, generator=g), dt)
    yb = 0.3 * torch.randn(cols, generator=g) if with_bias else None
    w, b = 1 + 0.2 * torch.randn(cols, generator=g), 0.1 * torch.randn(cols, generator=g)
    dh, ds = _r(torch.randn(rows, cols, generator=g), dt), _r(torch.randn(rows, cols, generator=g), dt)
    leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (res, y, yb, w, b)]
    ro, yo, ybo, wo, bo = leaves
    so = ro + yo + (ybo if with_bias else 0)
    if keep_sum and dt == torch.bfloat16:      # the kernel normalises the rounded sum it stores
        so = so + (_r(so.detach(), dt) - so.detach())
    ho = O.layer_norm(so, wo, bo, eps)
    (ho * dh).sum().backward(retain_graph=True) if not keep_sum else ((ho * dh).sum() + (so * ds).sum()).backward()
    dev = [t.to(DEV, dt if i < 2 else torch.float32).requires_grad_(True) if t is not None else None
           for i, t in enumerate((res, y, yb, w, b))]
    s, h = ops.add_layer_norm(dev[0], dev[1], dev[2], dev[3], dev[4], eps, keep_sum=keep_sum)
    assert (s is None) == (not keep_sum)
    loss = (h.float() * dh.to(DEV)).sum()
    if keep_sum:
        loss = loss + (s.float() * ds.to(DEV)).sum()
        _close(s, so.detach(), dt, 4, 's')
    loss.backward()
    _close(h, ho.detach(), dt, 6, 'h')
    _close(dev[0].grad, ro.grad, dt, 6, 'dres')
    _close(dev[1].grad, yo.grad, dt, 6, 'dy')
    if with_bias:
        _close(dev[2].grad, ybo.grad, dt, 10, 'dybias')
    _close(dev[3].grad, wo.grad, dt, 8, 'dgamma')
    _close(dev[4].grad, bo.grad, dt, 8, 'dbeta')


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('with_bias,second_use', [(True, True), (False, True), (True, False)])
def test_add_layernorm_pass_through(dt, with_bias, second_use):
    """h = LN(res + y + b) with res handed through for a second consumer (the x + space_out use of SpaceTimeBlock):
    d res = LN-backward + gradient of the second use, d y and d b see the LN-backward only."""
    from lavila_amd import ops
    rows, cols, eps = 93, 768, 1e-6
    g = torch.Generator().manual_seed(17)
    res, y = _r(torch.randn(rows, cols, generator=g), dt), _r(torch.randn(rows, cols, generator=g), dt)
    yb = 0.3 * torch.randn(cols, generator=g) if with_bias else None
    w, b = 1 + 0.2 * torch.randn(cols, generator=g), 0.1 * torch.randn(cols, generator=g)
    dh, d2 = _r(torch.randn(rows, cols, generator=g), dt), _r(torch.randn(rows, cols, generator=g), dt)
    ro, yo, ybo, wo, bo = [t.clone().requires_grad_(True) if t is not None else None for t in (res, y, yb, w, b)]
    ho = O.layer_norm(ro + yo + (ybo if with_bias else 0), wo, bo, eps)
    ((ho * dh).sum() + ((ro * d2).sum() if second_use else 0)).backward()
    dev = [t.to(DEV, dt if i < 2 else torch.float32).requires_grad_(True) if t is not None else None
           for i, t in enumerate((res, y, yb, w, b))]
    r2, h = ops.add_layer_norm_pass(dev[0], dev[1], dev[2], dev[3], dev[4], eps)
    assert torch.equal(r2, dev[0])
    loss = (h.float() * dh.to(DEV)).sum()
    if second_use:
        loss = loss + (r2.float() * d2.to(DEV)).sum()
    loss.backward()
    _close(h, ho.detach(), dt, 6, 'h')
    _close(dev[0].grad, ro.grad, dt, 6, 'dres')
    _close(dev[1].grad, yo.grad, dt, 6, 'dy')
    if with_bias:
        _close(dev[2].grad, ybo.grad, dt, 10, 'dybias')
    _close(dev[3].grad, wo.grad, dt, 8, 'dgamma')
    _close(dev[4].grad, bo.grad, dt, 8, 'dbeta')


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('rows,cols,with_bias', [(33, 3072, True), (5, 512, True), (300, 4096, True), (9, 2048, False)])
def test_bias_quickgelu(dt, rows, cols, with_bias):
    from lavila_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    u = _r(2 * torch.randn(rows, cols, generator=g), dt)
    b = 0.5 * torch.randn(cols, generator=g) if with_bias else None
    da = _r(torch.randn(rows, cols, generator=g), dt)
    uo = u.clone().requires_grad_(True)
    bo = b.clone().requires_grad_(True) if with_bias else None
    ao = O.quick_gelu(uo + bo if with_bias else uo)
    ao.backward(da)
    ug = u.to(DEV, dt).requires_grad_(True)
    bgp = b.to(DEV).requires_grad_(True) if with_bias else None
    a = ops.bias_quick_gelu(ug, bgp)
    a.backward(da.to(DEV, dt))
    _close(a, ao.detach(), dt, 4, 'a')
    _close(ug.grad, uo.grad, dt, 4, 'du')
    if with_bias:
        _close(bgp.grad, bo.grad, dt, 0.1 * rows + 1, 'dbias')


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,Fr,img,P', [(2, 2, 32, 16), (3, 4, 224, 16), (2, 3, 42, 14), (1, 2, 224, 14), (2, 1, 24, 8)])
def test_patchify_and_patch_embed(dt, B, Fr, img, P):
    from lavila_amd import ops
    g = torch.Generator().manual_seed(B + img)
    video = torch.randn(B, 3, Fr, img, img, generator=g)
    want = _r(O.patchify(video, P), dt)
    got = ops.patchify(video.to(DEV), P, dt)
    assert got.shape == want.shape
    assert torch.equal(got.float().cpu(), want)          # pure gather + one rounding: exact


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,Fr,N,D,num_frames', [(3, 2, 5, 128, 4), (5, 4, 196, 768, 4), (2, 1, 7, 64, 1), (9, 3, 33, 256, 8)])
def test_embed_tokens(dt, B, Fr, N, D, num_frames):
    """Forward (cls concat + positional / temporal add) and, round 5, the backward on lvl_embed_tokens_bwd (one pass over dx:
    d pos_embed, d temporal_embed incl. zero rows beyond the clip's frames, d cls_token) against autograd of the oracle."""
    from lavila_amd import ops
    g = torch.Generator().manual_seed(5)
    pe = _r(torch.randn(B, Fr * N, D, generator=g), dt)
    cls, pos, tem = (0.5 * torch.randn(s, generator=g) for s in ((1, 1, D), (1, N + 1, D), (1, num_frames, D)))
    dx = _r(torch.randn(B, 1 + Fr * N, D, generator=g), dt)
    lo = [t.clone().requires_grad_(True) for t in (pe, cls, pos, tem)]
    xo = torch.cat([lo[1].expand(B, -1, -1), lo[0]], 1) + O.total_pos_embed(lo[2], lo[3], N, Fr)
    xo.backward(dx)
    lg = [t.to(DEV, dt if i == 0 else torch.float32).requires_grad_(True) for i, t in enumerate((pe, cls, pos, tem))]
    x = ops.embed_tokens(lg[0], lg[1], lg[2], lg[3], Fr, N)
    x.backward(dx.to(DEV, dt))
    _close(x, xo.detach(), dt, 2, 'x')
    for a, b, n in zip(lg, lo, ('dpe', 'dcls', 'dpos', 'dtemporal')):
        assert a.grad.shape == b.grad.shape, n
        _close(a.grad, b.grad, dt, 4, n)


def _attn_case(B, Fr, N, H, seed):
    g = torch.Generator().manual_seed(seed)
    T, D = 1 + Fr * N, 64 * H
    return torch.randn(B, T, 3 * D, generator=g) * 1.5, torch.randn(B, T, D, generator=g)


def _check_qkv_bias_grad(got, dqkv_oracle, dt):
    """d(bias) of the qkv Linear = column sums of the oracle's dqkv. The product derives the v third from dout
    (softmax rows sum to 1) and sets the k third to its exact value 0 (softmax is shift-invariant in the keys)."""
    want = dqkv_oracle.sum((0, 1))
    D = want.numel() // 3
    scale = want.abs().max().item() + 1e-6
    tol = (3e-2 if dt == torch.bfloat16 else 2e-5) * scale
    assert (got.cpu() - want).abs().max().item() < tol, ((got.cpu() - want).abs().max().item(), tol)
    assert want[D:2 * D].abs().max().item() < 1e-4 * scale + 1e-5          # the oracle's own k third: rounding noise
    assert float(got[D:2 * D].abs().max()) == 0.0


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('mode', ['space', 'time'])
@pytest.mark.parametrize('B,Fr,N,H', [(2, 3, 5, 2), (2, 2, 49, 3), (3, 1, 7, 2), (2, 16, 4, 1), (1, 4, 196, 2), (1, 2, 256, 12), (1, 4, 196, 16), (2, 8, 33, 3),
                                      (2, 4, 70, 1), (1, 16, 10, 12), (1, 8, 20, 16), (2, 16, 33, 2), (1, 2, 576, 2), (1, 1, 400, 3),
                                      (1, 2, 591, 1), (1, 1, 272, 1), (2, 12, 7, 4), (1, 5, 9, 4), (1, 16, 196, 4), (2, 7, 3, 8),
                                      # 32-key block boundaries of the fused space backward (cls query = query row N)
                                      (2, 2, 1, 2), (1, 3, 31, 2), (2, 2, 32, 1), (1, 2, 63, 2), (1, 1, 64, 3), (1, 2, 287, 1),
                                      (1, 1, 288, 1)])
def test_divided_attention_core(dt, mode, B, Fr, N, H):
    from lavila_amd import ops
    qkv, dout = _attn_case(B, Fr, N, H, 17 + Fr + N)
    qkv, dout = _r(qkv, dt), _r(dout, dt)
    qo = qkv.clone().requires_grad_(True)
    oo = O.divided_attention_core(qo, H, Fr, N, mode)
    oo.backward(dout)
    qg = qkv.to(DEV, dt).requires_grad_(True)
    bias = torch.zeros(qkv.shape[-1], device=DEV, requires_grad=True)     # stands for the qkv Linear's bias
    o = ops.divided_attention(qg, Fr, N, H, mode, bias=bias)
    o.backward(dout.to(DEV, dt))
    _close(o, oo.detach(), dt, 2, 'out')
    _close(qg.grad, qo.grad, dt, 6, 'dqkv')
    _check_qkv_bias_grad(bias.grad, qo.grad, dt)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('mode', ['space', 'time'])
@pytest.mark.parametrize('B,Fr,N,H', [(3, 4, 196, 12), (2, 16, 49, 4), (2, 2, 400, 2), (2, 3, 5, 2)])
def test_divided_attention_backward_is_bitwise_reproducible(dt, mode, B, Fr, N, H):
    """Round 6: the cls token's d(q | k | v) is the sum of one partial record per frame (space kernels) / per location chunk
    (time kernels), added up in slot order by cls_grad_finalize_kernel -- no floating-point atomics any more. With f32
    atomicAdd on one record the order of 4..25 additions depended on timing; at the TSF-B geometry (196 locations, 25
    chunks) that flipped a bf16 rounding of dqkv's cls row about one step in six and the training step had two
    outcomes (profiles/r06_second_outcome.txt). Ten backward passes must agree to the bit, the bias gradient included;
    shapes: the fused space / register time kernels (TSF-B), the MFMA time kernels (16 frames), the streaming space
    kernels (400 locations), tiny groups."""
    from lavila_amd import ops
    qkv, dout = _attn_case(B, Fr, N, H, 5)
    qg = qkv.to(DEV, dt).requires_grad_(True)
    bias = torch.zeros(qkv.shape[-1], device=DEV, requires_grad=True)
    go = dout.to(DEV, dt)
    first = None
    for _ in range(10):
        qg.grad = bias.grad = None
        ops.divided_attention(qg, Fr, N, H, mode, bias=bias).backward(go)
        got = (qg.grad.clone(), bias.grad.clone())
        if first is None:
            first = got
        assert torch.equal(got[0], first[0]) and torch.equal(got[1], first[1])
    assert bool(torch.isfinite(first[0].float()).all())


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,L,ctx,V,W', [(5, 7, 77, 64, 512), (64, 72, 77, 512, 256), (256, 32, 77, 49408, 512), (3, 77, 77, 49408, 768),
                                         (1, 1, 4, 3, 8)])
def test_text_embedding_matches_nn_embedding(dt, B, L, ctx, V, W):
    """lvl_text_embed_fwd / _bwd against `token_embedding(text) + positional_embedding` (models.py:152-153) under autograd:
    the forward equals torch's to the bit (one f32 add, one rounding), d(table) / d(positional_embedding) equal float64
    sums of the same dx rows to float32 rounding, a second backward is bit-identical (no float atomics, no sort), and the
    token tensor is read through a strided `text[:, :L]` view. Tokens are drawn from a small range so that most rows have
    duplicates (the case the reduction exists for), the larger cases add a padding id with thousands of duplicates; (64, 72) is
    above torch's 3072-row switch to the rocPRIM sort path."""
    from lavila_amd import ops
    g = torch.Generator().manual_seed(3 + B + L)
    full = torch.randint(0, min(V, 97), (B, ctx), generator=g)
    full[:, 0] = V - 1
    if B >= 64:                          # ragged captions: the padding id fills a quarter of the positions (thousands of duplicates)
        full[:, (3 * L) // 4:] = 0
    table = (torch.randn(V, W, generator=g) * 0.1).to(DEV).requires_grad_(True)
    pos = (torch.randn(ctx, W, generator=g) * 0.1).to(DEV).requires_grad_(True)
    text = full.to(DEV)[:, :L]
    up = torch.randn(B, L, W, generator=g).to(DEV)
    x = ops.text_embed(text, table, pos, dt)
    assert x is not None and x.dtype == dt
    want = (torch.nn.functional.embedding(text, table) + pos[:L]).to(dt)
    assert torch.equal(x, want)
    (x.float() * up).sum().backward()
    dt1, dp1 = table.grad.clone(), pos.grad.clone()
    table.grad = pos.grad = None
    (ops.text_embed(text, table, pos, dt).float() * up).sum().backward()
    assert torch.equal(table.grad, dt1) and torch.equal(pos.grad, dp1)
    dx = up.to(dt).double()                                    # what the backward kernel reads: dx rounded to dt
    ref_t = torch.zeros(V, W, dtype=torch.float64, device=DEV).index_add_(0, text.reshape(-1), dx.reshape(-1, W))
    ref_p = torch.zeros(ctx, W, dtype=torch.float64, device=DEV)
    ref_p[:L] = dx.sum(0)
    assert (dt1.double() - ref_t).abs().max().item() <= 1e-6 * max(1.0, ref_t.abs().max().item())
    assert (dp1.double() - ref_p).abs().max().item() <= 1e-6 * max(1.0, ref_p.abs().max().item())
    assert float(dp1[L:].abs().max() if L < ctx else 0.0) == 0.0


@pytest.mark.parametrize('mode', ['space', 'time'])
@pytest.mark.parametrize('site', ['residual_epilogue', 'add_layernorm_pass'])
def test_colsum_tokens_give_the_bias_gradient_without_reading_dout(mode, site, monkeypatch):
    """Round 5: the v third of d(qkv bias) = sum_rows(dout) travels as a column-sum TOKEN from the consumer of the attention
    output back to the attention backward (sum_rows(dout) = sum_rows(dy) . W_proj, lvl_vec_mat_f32) instead of a pass over
    dout. Both consumer sites of SpaceTimeBlock.chain -- projection with the residual epilogue + LayerNorm (space), bias-free
    projection + fused add / LayerNorm with the residual handed through (time) -- with tokens on and off: the q / k thirds
    are bit-identical (same kernels), the v third agrees to the rounding of the bf16 dout rows the off-path sums, and both
    agree with float64 column sums of the float32 oracle's dqkv."""
    from lavila_amd import ops
    B, Fr, N, H = 2, 4, 49, 4
    D, T = 64 * H, 1 + Fr * N
    g = torch.Generator().manual_seed(11)
    qkv = (torch.randn(B, T, 3 * D, generator=g) * 1.2).to(torch.bfloat16)
    W = torch.randn(D, D, generator=g) * D ** -0.5
    pb, gam, bet = 0.1 * torch.randn(D, generator=g), 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    res = torch.randn(B, T, D, generator=g).to(torch.bfloat16)
    up, up2 = torch.randn(B, T, D, generator=g), torch.randn(B, T, D, generator=g)

    def run(tokens):
        monkeypatch.setattr(ops, 'COLSUM_TOKENS', tokens)
        q = qkv.to(DEV).requires_grad_(True)
        bias = torch.zeros(3 * D, device=DEV, requires_grad=True)
        Wd = W.to(DEV).requires_grad_(True)
        o, tok = ops.divided_attention(q, Fr, N, H, mode, bias=bias, want_token=True)
        assert (tok is not None) == tokens
        if site == 'residual_epilogue':
            s_, h = ops.linear_residual_layer_norm(o, Wd, pb.to(DEV), res.to(DEV), gam.to(DEV), bet.to(DEV), 1e-6, xtoken=tok)
        else:
            (y, ty) = ops.linear_with_token(o, Wd, tok)
            s_, h = ops.add_layer_norm_pass(res.to(DEV).requires_grad_(True), y, pb.to(DEV).requires_grad_(True), gam.to(DEV),
                                            bet.to(DEV), 1e-6, ytoken=ty)
        ((h.float() * up.to(DEV)).sum() + (s_.float() * up2.to(DEV)).sum()).backward()
        torch.cuda.synchronize()
        return bias.grad.clone(), q.grad.clone()

    (b_on, dq_on), (b_off, dq_off) = run(True), run(False)
    assert torch.equal(dq_on, dq_off)
    assert torch.equal(b_on[:2 * D], b_off[:2 * D]) and float(b_on[D:2 * D].abs().max()) == 0.0
    scale = b_off[2 * D:].abs().max().item()
    assert (b_on[2 * D:] - b_off[2 * D:]).abs().max().item() < 2e-2 * scale
    want = dq_on.double().sum((0, 1)).cpu()                     # column sums of the dqkv the kernels wrote (bf16 rows)
    assert (b_on.double().cpu() - want).abs().max().item() < 3e-2 * want.abs().max().item()


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,L,H', [(3, 77, 2), (2, 5, 1), (2, 130, 2)])
def test_causal_attention_core(dt, B, L, H):
    from lavila_amd import ops
    g = torch.Generator().manual_seed(L)
    qkv = _r(torch.randn(B, L, 3 * 64 * H, generator=g) * 1.5, dt)
    dout = _r(torch.randn(B, L, 64 * H, generator=g), dt)
    qo = qkv.clone().requires_grad_(True)
    oo = O.causal_attention_core(qo, H)
    oo.backward(dout)
    qg = qkv.to(DEV, dt).requires_grad_(True)
    bias = torch.zeros(qkv.shape[-1], device=DEV, requires_grad=True)
    o = ops.causal_attention(qg, H, bias=bias)
    o.backward(dout.to(DEV, dt))
    _close(o, oo.detach(), dt, 2, 'out')
    _close(qg.grad, qo.grad, dt, 6, 'dqkv')
    _check_qkv_bias_grad(bias.grad, qo.grad, dt)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,T,H', [(2, 99, 3), (3, 785, 12), (1, 1, 1), (2, 33, 2), (2, 3137, 12)])
def test_cls_attention_core(dt, B, T, H):
    """lvl_cls_attn_fwd / _bwd (the cls query of the last block's space attention over all T tokens) against
    oracle.cls_attention_core and its autograd gradients; the qkv bias gradient from the softmax identities."""
    from lavila_amd import ops
    D = 64 * H
    g = torch.Generator().manual_seed(B * 7 + T)
    q = _r(torch.randn(B, D, generator=g), dt)
    kv = _r(torch.randn(B, T, 2 * D, generator=g), dt)
    dout = _r(torch.randn(B, D, generator=g), dt)
    bias = torch.zeros(3 * D, device=DEV, requires_grad=True)
    qo, kvo = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    want = O.cls_attention_core(qo, kvo, H)
    want.backward(dout)
    qd, kvd = q.to(DEV).to(dt).requires_grad_(True), kv.to(DEV).to(dt).requires_grad_(True)
    got = ops.cls_attention(qd, kvd, H, bias=bias)
    got.backward(dout.to(DEV).to(dt))
    _close(got.detach().float().cpu(), want.detach(), dt, msg='out')
    _close(qd.grad.float().cpu(), qo.grad, dt, scale=max(1.0, qo.grad.abs().max().item()), msg='dq')
    _close(kvd.grad.float().cpu(), kvo.grad, dt, scale=max(1.0, kvo.grad.abs().max().item()), msg='dkv')
    # d bias = (sum_b dq | sum_{b,j} dk = 0 | sum_{b,j} dv): what adding the bias to q / k / v implies
    wantb = torch.cat([qo.grad.sum(0), kvo.grad[..., :D].sum((0, 1)), kvo.grad[..., D:].sum((0, 1))])
    tolb = (1e-3 if dt == torch.float32 else 5e-2) * max(1.0, wantb.abs().max().item())
    assert (bias.grad.cpu() - wantb).abs().max() < tolb
    assert float(bias.grad[D:2 * D].abs().max()) == 0.0


@pytest.mark.parametrize('case', range(4))
@pytest.mark.parametrize('mode', ['space', 'time'])
def test_var_attention_module_vs_reference_golden(case, mode):
    """Whole VarAttention layer (qkv Linear -> HIP core -> proj Linear) against outputs of the reference."""
    from lavila.models.timesformer import VarAttention
    rec = load_golden('var_attention.pt')[case]
    D = 64 * rec['H']
    m = VarAttention(D, num_heads=rec['H'], qkv_bias=True)
    m.load_state_dict(O.procedural_weights(rec['shapes'], seed=11))
    m.to(DEV)
    x = rec['x'].to(DEV).requires_grad_(True)
    pat = {'space': ('b (f n) d', '(b f) n d', {'f': rec['F']}), 'time': ('b (f n) d', '(b n) f d', {'n': rec['N']})}[mode]
    y = m(x, *pat)
    y.backward(rec['gout'].to(DEV))
    torch.testing.assert_close(y.detach().cpu(), rec[mode]['y'], atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(x.grad.cpu(), rec[mode]['dx'], atol=1e-4, rtol=1e-3)
    for k, p in m.named_parameters():
        torch.testing.assert_close(p.grad.cpu(), rec[mode]['dw'][k], atol=3e-4, rtol=1e-3, msg=lambda s: f'{k}: {s}')


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,G,E,row0', [(4, 4, 64, 0), (3, 12, 32, 6), (32, 256, 256, 64), (5, 300, 8, 295)])
def test_clip_loss_slabs(dt, B, G, E, row0):
    from helpers import oracle_slab_backward, oracle_slab_forward
    from lavila_amd import ops
    g = torch.Generator().manual_seed(G + E)
    img = _r(O.l2_normalize(torch.randn(G, E, generator=g)), dt)
    txt = _r(O.l2_normalize(torch.randn(G, E, generator=g) + 0.5 * img), dt)
    scale = torch.tensor([14.285714])
    lse_all = torch.stack([torch.logsumexp(O.clip_logits(img, txt, scale[0]), 1),
                           torch.logsumexp(O.clip_logits(img, txt, scale[0]), 0)])
    st_o, am_o = oracle_slab_forward(img, txt, scale[0], B, row0)
    up = torch.tensor([0.7])
    di_o, dt_o = oracle_slab_backward(img, txt, lse_all, scale, up, 3.0 / (2 * G), B, row0)
    ig, tg = img.to(DEV, dt), txt.to(DEV, dt)
    st, am, lg = ops.clip_loss_fwd_raw(ig, tg, scale.to(DEV), B, row0, want_logits=True)
    li = O.clip_logits(img, txt, scale[0])
    want_logits = torch.stack([li[row0:row0 + B], li.t()[row0:row0 + B]])
    torch.testing.assert_close(lg.cpu(), want_logits, atol=1e-4, rtol=1e-4)       # f32 accumulate in both dtypes
    torch.testing.assert_close(st.cpu(), st_o, atol=1e-4, rtol=1e-4)
    assert torch.equal(am.cpu(), am_o)                                             # indices: bit-exact
    di, dtx = ops.clip_loss_bwd_raw(ig, tg, lse_all.to(DEV), scale.to(DEV), up.to(DEV), 3.0 / (2 * G), B, row0)
    torch.testing.assert_close(di.cpu(), di_o, atol=2e-5, rtol=1e-3)
    torch.testing.assert_close(dtx.cpu(), dt_o, atol=2e-5, rtol=1e-3)
    # rows_only (local_loss without gather_with_grad): only the local entries of lse_all may be read
    lse_loc = torch.full_like(lse_all, float('nan'))
    lse_loc[:, row0:row0 + B] = lse_all[:, row0:row0 + B]
    di_o, dt_o = oracle_slab_backward(img, txt, lse_all, scale, up, 1.0 / (2 * B), B, row0, rows_only=True)
    di, dtx = ops.clip_loss_bwd_raw(ig, tg, lse_loc.to(DEV), scale.to(DEV), up.to(DEV), 1.0 / (2 * B), B, row0,
                                    rows_only=True)
    torch.testing.assert_close(di.cpu(), di_o, atol=2e-5, rtol=1e-3)
    torch.testing.assert_close(dtx.cpu(), dt_o, atol=2e-5, rtol=1e-3)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,G,E,row0', [(4, 4, 64, 0), (3, 12, 32, 6), (32, 256, 256, 64), (5, 300, 8, 295)])
def test_ssl_clip_loss_slabs(dt, B, G, E, row0):
    """lvl_ssl_clip_loss_fwd/bwd (per-pair temperature InfoNCE, loss.py:121-217) against the oracle."""
    from helpers import oracle_ssl_slab_backward, oracle_ssl_slab_forward
    from lavila_amd import ops
    img, txt, ind = O.ssl_synthetic_inputs(G, E, G + E)
    img, txt = _r(img, dt), _r(txt, dt)
    real, pseudo = torch.tensor(14.285714), torch.tensor(12.5)
    scales3 = torch.stack([pseudo, torch.sqrt(pseudo * real), real])
    li = O.ssl_scale_matrix(ind, real, pseudo) * (img @ txt.t())
    lse_all = torch.stack([torch.logsumexp(li, 1), torch.logsumexp(li, 0)])
    st_o, am_o = oracle_ssl_slab_forward(img, txt, ind, scales3, B, row0)
    up = torch.tensor([0.7])
    di_o, dt_o = oracle_ssl_slab_backward(img, txt, ind, lse_all, scales3, up, 3.0 / (2 * G), B, row0)
    ig, tg, ing = img.to(DEV, dt), txt.to(DEV, dt), ind.to(DEV, torch.int32)
    st, am, lg = ops.ssl_clip_loss_fwd_raw(ig, tg, ing, scales3.to(DEV), B, row0, want_logits=True)
    want_logits = torch.stack([li[row0:row0 + B], li.t()[row0:row0 + B]])
    torch.testing.assert_close(lg.cpu(), want_logits, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(st.cpu()[..., :7], st_o[..., :7], atol=1e-4, rtol=1e-4)
    assert torch.equal(am.cpu(), am_o)
    di, dtx = ops.ssl_clip_loss_bwd_raw(ig, tg, ing, lse_all.to(DEV), scales3.to(DEV), up.to(DEV), 3.0 / (2 * G), B,
                                        row0)
    torch.testing.assert_close(di.cpu(), di_o, atol=2e-5, rtol=1e-3)
    torch.testing.assert_close(dtx.cpu(), dt_o, atol=2e-5, rtol=1e-3)


def test_ssl_clip_loss_module_matches_reference_golden():
    """SSLCLIPLoss on the GPU through the C ABI against the reference's own outputs (tests/golden/ssl_clip_loss.pt)."""
    from conftest import load_golden
    from lavila.models.loss import SSLCLIPLoss
    fx = load_golden('ssl_clip_loss.pt')
    want = fx['single']
    img, txt, ind = O.ssl_synthetic_inputs(fx['single_G'], fx['E'], fx['seed'])
    li, lt = img.to(DEV).requires_grad_(True), txt.to(DEV).requires_grad_(True)
    scale = torch.tensor(fx['scale'], device=DEV).requires_grad_(True)
    crit = SSLCLIPLoss(scale_init=fx['scale_init']).to(DEV)
    out = crit({'image_embed': li, 'text_embed': lt, 'logit_scale': scale}, ind)       # indicators arrive on the CPU
    out['loss'].backward()
    for k in ('loss', 'clip_loss', 'clip_acc', 'clip_acc_gt', 'clip_acc_pseudo', 'num_gt', 'num_pseudo'):
        assert abs(float(out[k]) - want['out'][k]) < 1e-4, (k, float(out[k]), want['out'][k])
    assert abs(scale.grad.item() - want['dscale']) < 1e-5
    assert abs(crit.logit_scale_pseudo.grad.item() - want['dpseudo_param']) < 1e-5
    torch.testing.assert_close(li.grad.cpu(), want['dimg'], atol=1e-6, rtol=1e-4)
    torch.testing.assert_close(lt.grad.cpu(), want['dtxt'], atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize('M,N,K', [(4096, 768, 768), (1000, 2304, 768), (2053, 3072, 768), (4100, 768, 3072),
                                   (300, 576, 576), (33, 768, 768), (3000, 1024, 1024), (2100, 1536, 512),
                                   (1500, 512, 2048), (777, 512, 512), (640, 4096, 1024), (900, 256, 128),
                                   (500, 128, 256)])
def test_linear_wgrad_mfma(M, N, K):
    """lvl_linear_wgrad: dW = dY^T X and dbias = column sums of dY (bf16 operands, f32 accumulate) against a torch
    f32 reference on the same bf16-rounded inputs; ragged M (not a multiple of the 32-row step) included."""
    from lavila_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).bfloat16()
    x = torch.randn(M, K, generator=g).bfloat16()
    want = dy.float().t() @ x.float()
    dw, db = ops.linear_wgrad_raw(dy.to(DEV), x.to(DEV), True)
    torch.testing.assert_close(dw.cpu(), want, atol=2e-3, rtol=2e-5)
    torch.testing.assert_close(db.cpu(), dy.float().sum(0), atol=1e-3, rtol=1e-5)
    dw2, none = ops.linear_wgrad_raw(dy.to(DEV), x.to(DEV), False)
    assert none is None and torch.equal(dw2, dw)                     # deterministic (no atomics)


@pytest.mark.parametrize('rows,n_in,n_out', [(8192, 768, 2304), (6000, 3072, 768), (5000, 512, 1536), (300, 768, 768),
                                             (33000, 768, 800), (40000, 100, 72)])
def test_linear_layer_bf16_training_path(rows, n_in, n_out):
    """ops.linear as the towers call it under bf16 autocast: forward GEMM, input gradient against the transposed
    weight copy, weight gradient through lvl_linear_wgrad (>= 4096 rows) or the library (fewer), bias gradient --
    against an f32 torch reference on the same bf16-rounded operands."""
    from lavila_amd import ops
    g = torch.Generator().manual_seed(rows + n_in)
    x = torch.randn(rows, n_in, generator=g).bfloat16()
    w = (torch.randn(n_out, n_in, generator=g) * 0.03)
    b = torch.randn(n_out, generator=g) * 0.1
    dy = torch.randn(rows, n_out, generator=g).bfloat16()
    xr = x.float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    br = b.bfloat16().float().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(dy.float())
    xg = x.to(DEV).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)             # f32 master weights, cast per GEMM like autocast does
    bg = b.to(DEV).requires_grad_(True)
    y = ops.linear(xg, wg, bg)
    assert y.dtype == torch.bfloat16
    y.backward(dy.to(DEV))
    torch.testing.assert_close(y.float().cpu(), yr.detach(), atol=3e-2, rtol=2e-2)            # bf16 output rounding
    torch.testing.assert_close(xg.grad.float().cpu(), xr.grad, atol=3e-2, rtol=2e-2)
    scale = wr.grad.abs().max().item()
    assert wg.grad.dtype == torch.float32
    assert (wg.grad.cpu() - wr.grad).abs().max().item() < 4e-3 * scale + 1e-3               # f32 accumulation
    torch.testing.assert_close(bg.grad.cpu(), br.grad, atol=2e-2, rtol=1e-3)


def test_linear_wgrad_unsupported_shape_is_loud():
    from lavila_amd import ops
    from lavila_amd._cabi import HipExtensionError
    with pytest.raises(HipExtensionError):
        ops.linear_wgrad_raw(torch.zeros(64, 200, device=DEV, dtype=torch.bfloat16),
                             torch.zeros(64, 200, device=DEV, dtype=torch.bfloat16), False)


def test_kernel_argument_errors_are_loud():
    from lavila_amd import ops
    from lavila_amd._cabi import HipExtensionError
    x = torch.randn(4, 100, device=DEV)          # cols % 8 != 0
    with pytest.raises(HipExtensionError):
        ops.layer_norm(x, torch.ones(100, device=DEV), torch.zeros(100, device=DEV), 1e-5)
    with pytest.raises(HipExtensionError):
        ops.divided_attention(torch.randn(2, 11, 3 * 96, device=DEV), 2, 5, 3, 'space')   # head dim 32
    with pytest.raises(HipExtensionError):
        ops.layer_norm(torch.randn(4, 128, device=DEV, dtype=torch.float64), torch.ones(128, device=DEV),
                       torch.zeros(128, device=DEV), 1e-5)
    # fp16 activations (model.half() callers) are computed in bf16, not rejected (tests/test_gpu_boundary.py)
    xh = torch.randn(4, 128, device=DEV, dtype=torch.float16)
    yh = ops.layer_norm(xh, torch.ones(128, device=DEV), torch.zeros(128, device=DEV), 1e-5)
    assert yh.dtype == torch.bfloat16
    torch.testing.assert_close(yh.float(), torch.nn.functional.layer_norm(xh.float(), (128,)), atol=3e-2, rtol=3e-2)


def test_block_with_residual_epilogue_equals_the_composed_block():
    """ops.RESIDUAL_EPILOGUE: the space attention's projection adds the residual in its GEMM epilogue and norm2 reads the
    sum (ops._LinearResidualLayerNormFn) -- same block output and parameter / input gradients as projection + fused
    add + LayerNorm, to one bf16 rounding of x1 (the epilogue adds in f32 BEFORE rounding; the composed form rounds the
    projection's output first)."""
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeBlock
    from lavila_amd import ops
    torch.manual_seed(2)
    Fr, N, D, H, B = 4, 196, 768, 12, 2
    blk = SpaceTimeBlock(D, H, qkv_bias=True, act_layer=QuickGELU, time_init='rand').to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            if p.ndim > 1:
                p.normal_(0, 0.02)
            else:
                p.add_(torch.randn_like(p) * 0.05)
    x = torch.randn(B, 1 + Fr * N, D, device=DEV, dtype=torch.bfloat16)
    gout = torch.randn(B, 1 + Fr * N, D, device=DEV, dtype=torch.bfloat16)
    res = []
    shipped = ops.RESIDUAL_EPILOGUE
    for flag in (False, True):
        ops.RESIDUAL_EPILOGUE = flag
        try:
            for p in blk.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                x1, y, b = blk.chain(xi, None, None, Fr, N)
                out = x1 + y + b.to(y.dtype)
            out.backward(gout)
            res.append((out.detach().float(), xi.grad.float(), {n: p.grad.float().clone() for n, p in blk.named_parameters()}))
        finally:
            ops.RESIDUAL_EPILOGUE = shipped
    (o0, g0, p0), (o1, g1, p1) = res
    assert not torch.equal(o0, o1)          # the flag took the other path (x1 is rounded once instead of twice)
    torch.testing.assert_close(o1, o0, atol=6e-2, rtol=2e-2)
    assert ((o1 - o0).norm() / o0.norm()).item() < 4e-3
    assert ((g1 - g0).norm() / g0.norm()).item() < 1e-2
    for n in p0:
        d = (p1[n] - p0[n]).norm().item()
        assert d <= 1e-2 * p0[n].norm().item() + 1e-6, (n, d, p0[n].norm().item())


def test_tower_with_residual_epilogues_equals_the_composed_tower():
    """ops.RESIDUAL_EPILOGUE through a 3-block tower: the space projection's and (deferred to the next block's norm3:
    timesformer.PendingMlp) the MLP's residual adds ride in GEMM epilogues. Same features and gradients as the composed
    form up to the roundings that moved (the stream is rounded once per add instead of product-then-sum)."""
    import contextlib
    import io
    import torch.nn as nn
    from lavila.models.openai_model import QuickGELU
    from lavila.models.timesformer import SpaceTimeTransformer
    from lavila_amd import ops
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        vis = SpaceTimeTransformer(img_size=224, patch_size=16, embed_dim=768, depth=3, num_heads=12, num_frames=4,
                                   time_init='zeros', attention_style='frozen-in-time', ln_pre=True, act_layer=QuickGELU,
                                   is_tanh_gating=False)
    vis.head = vis.pre_logits = vis.fc = nn.Identity()
    vis = vis.to(DEV).train()
    with torch.no_grad():
        for n, p in vis.named_parameters():
            if p.ndim > 1 and 'timeattn' in n:
                p.normal_(0, 0.02)          # time attention is zero-initialised: make every branch carry signal
    video = torch.randn(2, 3, 4, 224, 224, device=DEV)
    res = []
    shipped = ops.RESIDUAL_EPILOGUE
    for flag, amp in ((False, False), (False, True), (True, True)):        # float32 reference, composed bf16, epilogue bf16
        ops.RESIDUAL_EPILOGUE = flag
        try:
            for p in vis.parameters():
                p.grad = None
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
                feat = vis(video)
            feat.float().square().sum().backward()
            res.append((feat.detach().float(), {n: p.grad.float().clone() for n, p in vis.named_parameters()
                                                if p.grad is not None}))
        finally:
            ops.RESIDUAL_EPILOGUE = shipped
    (fr, pr), (f0, p0), (f1, p1) = res
    assert set(p0) == set(p1) == set(pr)
    assert not torch.equal(f0, f1)                                         # the flag took the other path

    def err(f, p):
        tot = sum(v.norm().item() ** 2 for v in pr.values()) ** 0.5
        dif = sum((p[n] - pr[n]).norm().item() ** 2 for n in pr) ** 0.5
        return ((f - fr).norm() / fr.norm()).item(), dif / tot

    (ef0, eg0), (ef1, eg1) = err(f0, p0), err(f1, p1)
    # two bf16 evaluations of one function differ from each other by about their distance to float32; what the epilogue
    # form must not be is FURTHER from float32 than the composed form (it rounds the stream once per add instead of twice)
    assert ef1 < 1.25 * ef0 + 1e-4 and eg1 < 1.25 * eg0 + 1e-4, (ef0, ef1, eg0, eg1)
    for n in ('blocks.1.mlp.fc2.bias', 'blocks.0.mlp.fc2.bias', 'blocks.1.attn.proj.bias', 'blocks.2.norm3.weight'):
        d0 = (p0[n] - pr[n]).norm().item() / pr[n].norm().item()
        d1 = (p1[n] - pr[n]).norm().item() / pr[n].norm().item()
        assert d1 < 1.5 * d0 + 2e-3, (n, d0, d1)


_LN_EXACT_PROBE = r"""
import sys, torch
from lavila_amd import ops
torch.manual_seed(0)
cols = int(sys.argv[2])
x = torch.randn(3001, cols, device='cuda').bfloat16(); y = torch.randn_like(x)
g = torch.randn(cols, device='cuda'); b = torch.randn(cols, device='cuda'); yb = torch.randn(cols, device='cuda')
dy = torch.randn_like(x); dadd = torch.randn_like(x)
outs = []
h, _, mean, rstd = ops.layernorm_fwd_raw(x, None, None, g, b, 1e-5, False); outs += [h, mean, rstd]
h2, _, m2, r2 = ops.layernorm_fwd_raw(x, y, yb, g, b, 1e-5, False); outs += [h2, m2, r2]
h3, _, m3, r3 = ops.layernorm_fwd_raw(x, y, None, g, b, 1e-5, False); outs += [h3, m3, r3]
outs += [t for t in ops.layernorm_bwd_raw(dy, x, None, None, g, mean, rstd, None, False) if t is not None]
outs += [t for t in ops.layernorm_bwd_raw(dy, x, None, None, g, mean, rstd, dadd, True) if t is not None]
outs += [t for t in ops.layernorm_bwd_raw(dy, x, y, yb, g, m2, r2, None, True) if t is not None]
outs += [t for t in ops.layernorm_bwd_raw(dy, x, y, yb, g, m2, r2, dadd, True, True) if t is not None]
torch.save([o.cpu() for o in outs], sys.argv[1])
"""


@pytest.mark.parametrize('cols', [768, 512, 1024])
def test_exact_width_layernorm_kernels_equal_the_general_ones_bit_for_bit(cols, tmp_path):
    """lvl_layernorm_fwd / bwd take branch-free exact-width instantiations (ln_fwd_exact_kernel / ln_bwd_exact_kernel) at
    cols = VPL * 256 for the operand combinations of the training step, the general kernels otherwise -- and activation
    checkpointing (timesformer.py:173-187) may recompute a forward through the other one. Both families write their fused
    multiply-adds explicitly (no implicit contraction in layernorm.hip), so they agree to the bit: LAVILA_LN_EXACT=0 / 1."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for e in ('0', '1'):
        out = str(tmp_path / f'ln_{e}.pt')
        subprocess.run([sys.executable, '-c', _LN_EXACT_PROBE, out, str(cols)], check=True, cwd=root,
                       env=dict(os.environ, LAVILA_LN_EXACT=e, PYTHONPATH=root), timeout=600)
        res.append(torch.load(out))
    assert len(res[0]) == len(res[1]) and len(res[0]) >= 20
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), (i, (a.float() - b.float()).abs().max().item())


def test_cast_transpose_multi_equals_the_single_launches_and_refresh_follows_the_weights():
    """lvl_cast_transpose_multi (every Linear weight's bf16 copy + transposed copy in one launch, what autocast's per-step weight
    casts amount to: main_pretrain.py:491,520-533) against lvl_cast_transpose weight by weight, odd shapes included; and
    ops.refresh_weight_copies: after an out-of-band `.data` write and the next training forward's generation bump the cached
    pairs hold the new values (one launch), entries of dead parameters are skipped."""
    from lavila_amd import _cabi as C
    from lavila_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    shapes = [(768, 768), (2304, 768), (70, 130), (1, 64), (513, 65), (3072, 768), (64, 64)]
    ws = [torch.randn(n, k, device=DEV, generator=g) for n, k in shapes]
    rows, outs, tile0 = [], [], 0
    for w in ws:
        n, k = w.shape
        a = torch.empty(n, k, dtype=torch.bfloat16, device=DEV)
        b = torch.empty(k, n, dtype=torch.bfloat16, device=DEV)
        rows.append([w.data_ptr(), a.data_ptr(), b.data_ptr(), n | (k << 32), tile0])
        tile0 += ((n + 63) // 64) * ((k + 63) // 64)
        outs.append((a, b))
    table = torch.tensor(rows, dtype=torch.int64, device=DEV)
    C.check(C.lib().lvl_cast_transpose_multi(C.ptr(table), len(rows), tile0, C.stream_ptr()), 'lvl_cast_transpose_multi')
    for w, (a, b) in zip(ws, outs):
        assert torch.equal(a, w.bfloat16()) and torch.equal(b, w.t().contiguous().bfloat16())
    # the cache: first use casts lazily, the next training forward refreshes in one go
    p1 = torch.nn.Parameter(torch.randn(256, 128, device=DEV, generator=g))
    p2 = torch.nn.Parameter(torch.randn(512, 256, device=DEV, generator=g))
    with ops.model_forward():
        ops.weight_copies(p1), ops.weight_copies(p2)
    p1.data.mul_(2.0)                      # behind the version counter's back
    p2.data.add_(1.0)
    del_me = torch.nn.Parameter(torch.randn(64, 64, device=DEV, generator=g))
    with ops.model_forward():
        ops.weight_copies(del_me)
    del del_me
    was = ops.WEIGHT_REFRESH
    ops.WEIGHT_REFRESH = True              # (opt-in: LAVILA_WEIGHT_REFRESH=1)
    try:
        with ops.model_forward():          # generation bump + refresh (one launch for everything recently used)
            entries = [v for v in ops._copies.values() if v[5] is not None and any(v[5]() is q for q in (p1, p2))]
            assert len(entries) == 2 and all(v[4] == ops._generation for v in entries)
            a1, t1 = ops.weight_copies(p1)
            a2, t2 = ops.weight_copies(p2)
    finally:
        ops.WEIGHT_REFRESH = was
    assert torch.equal(a1, p1.detach().bfloat16()) and torch.equal(t1, p1.detach().t().contiguous().bfloat16())
    assert torch.equal(a2, p2.detach().bfloat16()) and torch.equal(t2, p2.detach().t().contiguous().bfloat16())
