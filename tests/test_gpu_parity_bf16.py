"""GPU (-m gpu): parity evidence for the BENCHMARKED configuration -- bf16 activations on the MFMA kernels
(attention, forward / input-gradient / weight-gradient GEMMs) at TSF-B width and depth -- which the float32 goldens
of test_gpu_model.py do not exercise (float32 dispatches to the shape-generic kernels).

Two kinds of evidence:
  * EXACT structural tests: inputs built so that every intermediate value is exactly representable (one-hot
    attention, small-integer operands). The expected output is then known bit for bit, so an indexing error (a key
    dropped from one frame, a CLS row attached to the wrong group, a transposed tile) cannot hide under a bf16
    tolerance;
  * a whole training step of CLIP_OPENAI_TIMESFORMER_BASE on 4x224^2 clips under bf16 autocast against the float32
    CPU oracle, every parameter gradient compared by relative L2, with the bound derived from the bf16 unit roundoff
    (see test_tsfb_bf16_training_step_vs_oracle_f32).
"""
import math

import pytest
import torch

from helpers import oracle_slab_forward
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


# --------------------------------------------------------------------------------------------------------------------
# exact structural tests of the attention kernels
# --------------------------------------------------------------------------------------------------------------------
def _pair_codes(T):
    """T distinct unordered pairs (a, b), a < b < 64: token j's key is 64*(e_a + e_b)."""
    pairs = [(a, b) for a in range(64) for b in range(a + 1, 64)]
    g = torch.Generator().manual_seed(5)
    perm = torch.randperm(len(pairs), generator=g)[:T]
    return torch.tensor(pairs)[perm]          # [T, 2]


def _one_hot_problem(B, H, T, allowed, seed):
    """qkv [B,T,3*H*64] bf16 such that query t of head h attends to exactly ONE key target[b,h,t] drawn from
    allowed(t): q_t = k_target, keys are 64*(e_a+e_b) with distinct pairs, so the scaled score is 1024 on the target,
    <= 512 elsewhere: exp(-512) underflows to 0 in float32 and the softmax is exactly one-hot. v and dout hold small
    integers: every dot product and every sum is exact in float32 whatever the summation order."""
    g = torch.Generator().manual_seed(seed)
    codes = _pair_codes(T)
    keys = torch.zeros(T, 64)
    keys[torch.arange(T), codes[:, 0]] = 64.0
    keys[torch.arange(T), codes[:, 1]] = 64.0
    target = torch.empty(B, H, T, dtype=torch.long)
    for t in range(T):
        cand = torch.tensor(allowed(t))
        target[:, :, t] = cand[torch.randint(len(cand), (B, H), generator=g)]
    q = keys[target]                                                    # [B,H,T,64]
    k = keys[None, None].expand(B, H, T, 64)
    v = torch.randint(-4, 5, (B, H, T, 64), generator=g).float()
    dout = torch.randint(-3, 4, (B, H, T, 64), generator=g).float()
    pack = lambda x: x.permute(0, 2, 1, 3).reshape(B, T, H * 64)        # noqa: E731  head-major inside a third
    qkv = torch.cat([pack(q), pack(k), pack(v)], -1).to(torch.bfloat16)
    out = torch.gather(v, 2, target[..., None].expand(B, H, T, 64))    # out[t] = v[target(t)]
    dv = torch.zeros(B, H, T, 64).scatter_add_(2, target[..., None].expand(B, H, T, 64), dout)
    return qkv, pack(dout).to(torch.bfloat16), pack(out), pack(dv), target


def _check_exact(qkv, dout, out_want, dv_want, run):
    D = out_want.shape[-1]
    x = qkv.to(DEV).requires_grad_(True)
    out = run(x)
    out.backward(dout.to(DEV))
    assert torch.equal(out.float().cpu(), out_want), 'forward: out[t] != v[target(t)] bit for bit'
    g = x.grad.float().cpu()
    assert torch.equal(g[..., 2 * D:], dv_want), 'backward: dv != scatter-add of dout over the targets'
    assert torch.count_nonzero(g[..., :2 * D]) == 0, 'backward: dq / dk must be exactly zero for a one-hot softmax'


@pytest.mark.parametrize('mode,B,Fr,N,H', [('space', 2, 4, 196, 12), ('time', 2, 4, 196, 12), ('space', 1, 2, 49, 3),
                                           ('time', 1, 16, 4, 2), ('time', 1, 8, 9, 2), ('space', 1, 1, 256, 16),
                                           ('space', 1, 2, 576, 2), ('space', 1, 1, 400, 2), ('space', 1, 3, 591, 1),
                                           ('time', 1, 16, 9, 4), ('time', 2, 8, 5, 4), ('time', 1, 12, 7, 8), ('time', 1, 5, 6, 4),
                                           ('time', 1, 16, 100, 12)])
def test_divided_attention_one_hot_exact(mode, B, Fr, N, H):
    """Every query picks one key of its group (cls | same frame | same location; cls query: any token): the bf16 MFMA /
    register-tiled kernels must reproduce v[target] and the scatter-added dv exactly (timesformer.py:110-140)."""
    from lavila_amd import ops
    T = 1 + Fr * N

    def allowed(t):
        if t == 0:
            return list(range(T))
        f, n = divmod(t - 1, N)
        if mode == 'space':
            return [0] + [1 + f * N + m for m in range(N)]
        return [0] + [1 + ff * N + n for ff in range(Fr)]
    qkv, dout, out_want, dv_want, _ = _one_hot_problem(B, H, T, allowed, seed=11)
    _check_exact(qkv, dout, out_want, dv_want, lambda x: ops.divided_attention(x, Fr, N, H, mode))


@pytest.mark.parametrize('B,L,H', [(3, 77, 8), (2, 32, 8), (2, 130, 12)])
def test_causal_attention_one_hot_exact(B, L, H):
    """Causal text attention (openai_model.py:196-198): query t may only pick a key j <= t."""
    from lavila_amd import ops
    qkv, dout, out_want, dv_want, _ = _one_hot_problem(B, H, L, lambda t: list(range(t + 1)), seed=13)
    _check_exact(qkv, dout, out_want, dv_want, lambda x: ops.causal_attention(x, H))


# --------------------------------------------------------------------------------------------------------------------
# exact tests of the MFMA GEMMs (small-integer operands: float32 accumulation is exact, so is the bf16 result)
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(1000, 768, 768), (70001, 256, 192), (513, 2304, 768)])
def test_linear_tn_exact_on_integer_operands(M, N, K):
    from lavila_amd import _cabi as C
    from lavila_amd import ops
    g = torch.Generator().manual_seed(M)
    x = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float() * (torch.rand(N, K, generator=g) < 0.08)    # sparse +-1
    b = torch.randint(-8, 9, (N,), generator=g).float()
    want = x @ w.t() + b                      # |values| <= 2*0.08*K + 8 stays far below 256: exact in bf16
    assert want.abs().max() < 256
    y = ops.linear_tn_raw(x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), b.to(DEV), C.EPI_BIAS)
    assert torch.equal(y.float().cpu(), want)


@pytest.mark.parametrize('M,N,K', [(1000, 768, 768), (70001, 256, 192), (513, 768, 3072), (1, 256, 64)])
def test_linear_tn_residual_epilogue_exact_on_integer_operands(M, N, K):
    """LVL_EPI_BIAS_RESIDUAL: y = x W^T + b + res (`x + attn(...)`, `x + mlp(...)`, timesformer.py:183-196) -- exact on small
    integers (tail tiles, one-row problems included), and with a None bias."""
    from lavila_amd import _cabi as C
    from lavila_amd import ops
    g = torch.Generator().manual_seed(M + 1)
    x = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float() * (torch.rand(N, K, generator=g) < 0.08)
    b = torch.randint(-8, 9, (N,), generator=g).float()
    res = torch.randint(-60, 61, (M, N), generator=g).float()
    want = x @ w.t() + b + res
    assert want.abs().max() < 256
    xd, wd, rd = x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), res.to(DEV).bfloat16()
    y = ops.linear_tn_raw(xd, wd, b.to(DEV), C.EPI_BIAS_RESIDUAL, aux_in=rd)
    assert torch.equal(y.float().cpu(), want)
    y0 = ops.linear_tn_raw(xd, wd, None, C.EPI_BIAS_RESIDUAL, aux_in=rd)
    assert torch.equal(y0.float().cpu(), x @ w.t() + res)
    # random operands: the sum is rounded ONCE (f32 accumulator + f32 residual), so it is at least as close to the
    # float64 result as rounding the product first and adding then
    xr = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    wr = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV).bfloat16()
    rr = (torch.randn(M, N, generator=g) * 4).to(DEV).bfloat16()
    ref = xr.double() @ wr.double().t() + b.to(DEV).double() + rr.double()
    got = ops.linear_tn_raw(xr, wr, b.to(DEV), C.EPI_BIAS_RESIDUAL, aux_in=rr).double()
    two = (ops.linear_tn_raw(xr, wr, b.to(DEV), C.EPI_BIAS).float() + rr.float()).bfloat16().double()
    assert (got - ref).abs().max() <= 2.0 ** -8 * ref.abs().max() + 1e-6
    assert (got - ref).norm() <= (two - ref).norm() * 1.0001


@pytest.mark.parametrize('M,N,K', [(1000, 768, 768), (70001, 256, 192), (513, 3072, 768), (1, 256, 64)])
def test_linear_tn_quickgelu_derivative_epilogues(M, N, K):
    """LVL_EPI_BIAS_QUICKGELU_DERIV / LVL_EPI_MUL_AUX_COLSUM (the training pair of Mlp.fc1 + QuickGELU, timesformer.py:52-54,
    and of its backward): y = u sigmoid(1.702 u) and aux_out = d quickgelu(u) from the f32 accumulator, each rounded once;
    y = acc * aux_in exact on small integers, column sums in f32 (tail tiles and one-row problems included)."""
    from lavila_amd import _cabi as C
    from lavila_amd import ops
    g = torch.Generator().manual_seed(M + 2)
    xr = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    wr = (torch.randn(N, K, generator=g) * 2 * K ** -0.5).to(DEV).bfloat16()
    b = torch.randn(N, generator=g).to(DEV)
    u = xr.double() @ wr.double().t() + b.double()
    sg = torch.sigmoid(1.702 * u)
    y, d = ops.linear_tn_raw(xr, wr, b, C.EPI_BIAS_QUICKGELU_DERIV)
    assert y.dtype == torch.bfloat16 and d.dtype == torch.bfloat16
    want_y, want_d = u * sg, sg * (1 + 1.702 * u * (1 - sg))
    # one bf16 rounding (2^-9 relative) of a value computed in f32 from an f32 accumulator (|u| <~ 10)
    assert (y.double() - want_y).abs().max() <= 2.0 ** -8 * want_y.abs().max() + 1e-6
    assert ((y.double() - want_y).abs() <= 2.0 ** -8 * want_y.abs() + 2e-5).all()
    assert ((d.double() - want_d).abs() <= 2.0 ** -8 * want_d.abs() + 2e-5).all()
    # against the pre-activation form of the same kernel: the activation there sees bf16(u)
    y1, u1 = ops.linear_tn_raw(xr, wr, b, C.EPI_BIAS_QUICKGELU)
    assert (u1.double() - u).abs().max() <= 2.0 ** -8 * u.abs().max()
    assert (y1.double() - y.double()).abs().max() <= 2.0 ** -6 * want_y.abs().max()
    # backward form, exact: integer operands, integer 'derivatives'
    x = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float() * (torch.rand(N, K, generator=g) < 0.08)
    a = torch.randint(-2, 3, (M, N), generator=g).float()
    want = (x @ w.t()) * a
    assert want.abs().max() < 256
    got, cs = ops.linear_tn_raw(x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), None, C.EPI_MUL_AUX_COLSUM, aux_in=a.to(DEV).bfloat16())
    assert torch.equal(got.float().cpu(), want)
    assert torch.equal(cs.cpu(), want.sum(0))             # integers < 2^24: exact in any order
    # the two backward forms agree on random data to the rounding of the stored derivative
    dy = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    du5, cs5 = ops.linear_tn_raw(dy, wr, None, C.EPI_MUL_AUX_COLSUM, aux_in=d)
    du2, cs2 = ops.linear_tn_raw(dy, wr, None, C.EPI_QUICKGELU_BWD, aux_in=u1)
    ref = (dy.double() @ wr.double().t()) * want_d
    scale = ref.abs().max()
    assert (du5.double() - ref).abs().max() <= 2.0 ** -7 * scale
    assert (du2.double() - ref).abs().max() <= 2.0 ** -6 * scale
    assert (cs5.double() - ref.sum(0)).abs().max() <= 2.0 ** -7 * ref.abs().sum(0).max() + 1e-3
    # no f32-class instantiation: refused loudly
    with pytest.raises(C.HipExtensionError):
        ops.linear_tn_raw(ops.split3(xr.float(), 0), ops.split3(wr.float(), 1), b, C.EPI_BIAS_QUICKGELU_DERIV, f32=True)


def test_quickgelu_derivative_pair_at_the_benched_size_by_properties():
    """BASELINE configs[1] size (M = 256 * 785 rows, fc1 768 -> 3072): size-independent properties of the training pair of
    epilogues. (1) the activation of LVL_EPI_BIAS_QUICKGELU_DERIV equals the pre-activation form's up to the rounding of u it
    skips; (2) the stored derivative lies in QuickGELU's derivative range [-0.1, 1.1]; (3) LVL_EPI_MUL_AUX_COLSUM: the column
    sums equal the column sums of its own result rows (a checksum of checksums; the kernel sums before the bf16 rounding);
    (4) linearity in the upstream gradient: du(2 dy) == 2 du(dy) bit for bit (power-of-two scaling is exact)."""
    from lavila_amd import _cabi as C
    from lavila_amd import ops
    M, K, N = 256 * 785, 768, 3072
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w1 = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    b1 = torch.randn(N, device=DEV, generator=g) * 0.1
    y, d = ops.linear_tn_raw(x, w1, b1, C.EPI_BIAS_QUICKGELU_DERIV)
    y1, u1 = ops.linear_tn_raw(x, w1, b1, C.EPI_BIAS_QUICKGELU)
    assert (y.float() - y1.float()).abs().max().item() <= 2.0 ** -6 * y1.float().abs().max().item()
    assert d.float().min().item() >= -0.11 and d.float().max().item() <= 1.11
    del y1, u1
    dy = torch.randn(M, K, device=DEV, generator=g).bfloat16()        # stands for the upstream gradient times W2 (any [M, K])
    wt = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    du, cs = ops.linear_tn_raw(dy, wt, None, C.EPI_MUL_AUX_COLSUM, aux_in=d)
    want = du.double().sum(0)
    bound = 2.0 ** -8 * du.double().abs().sum(0) + 1e-3
    assert ((cs.double() - want).abs() <= bound).all(), float(((cs.double() - want).abs() / bound).max())
    du2, cs2 = ops.linear_tn_raw((dy.float() * 2).bfloat16(), wt, None, C.EPI_MUL_AUX_COLSUM, aux_in=d)
    assert torch.equal(du2.float(), du.float() * 2) and torch.equal(cs2, cs * 2)


@pytest.mark.parametrize('M,N,K', [(4096, 768, 768), (20000, 384, 192), (33, 2304, 768)])
def test_linear_wgrad_exact_on_integer_operands(M, N, K):
    from lavila_amd import ops
    g = torch.Generator().manual_seed(M + 1)
    dy = torch.randint(-2, 3, (M, N), generator=g).float() * (torch.rand(M, N, generator=g) < 0.05)
    x = torch.randint(-2, 3, (M, K), generator=g).float()
    want = dy.t() @ x                         # integers < 2^24: exact in float32 in any summation order
    dw, _ = ops.linear_wgrad_raw(dy.to(DEV).bfloat16(), x.to(DEV).bfloat16(), False)
    assert torch.equal(dw.cpu(), want)


@pytest.mark.parametrize('cus', [240, 64, 8])
def test_persistent_gemms_exact_with_fewer_compute_units(cus):
    """lvl_set_compute_units: the persistent kernels size their grids (and the weight gradient its row splits and
    workspace) for fewer CUs; the results stay exact."""
    from lavila_amd import _cabi as C
    from lavila_amd import ops
    lib = C.lib()
    assert lib.lvl_set_compute_units(12) != 0 and lib.lvl_set_compute_units(-8) != 0      # not a multiple of 8 / negative
    assert lib.lvl_set_compute_units(cus) == 0
    try:
        g = torch.Generator().manual_seed(cus)
        M, N, K = 9000, 768, 768
        x = torch.randint(-2, 3, (M, K), generator=g).float()
        w = torch.randint(-1, 2, (N, K), generator=g).float() * (torch.rand(N, K, generator=g) < 0.08)
        b = torch.randint(-8, 9, (N,), generator=g).float()
        y = ops.linear_tn_raw(x.to(DEV).bfloat16(), w.to(DEV).bfloat16(), b.to(DEV), C.EPI_BIAS)
        assert torch.equal(y.float().cpu(), x @ w.t() + b)
        dy = torch.randint(-2, 3, (M, N), generator=g).float() * (torch.rand(M, N, generator=g) < 0.05)
        dw, _ = ops.linear_wgrad_raw(dy.to(DEV).bfloat16(), x.to(DEV).bfloat16(), False)
        assert torch.equal(dw.cpu(), dy.t() @ x)
    finally:
        assert lib.lvl_set_compute_units(0) == 0


def _tn_call(x, w, bias, epi, sched, aux_in=None):
    """lvl_linear_tn through the raw C ABI with an explicit tile-counter block (None = static tile ranges)."""
    from lavila_amd import _cabi as C
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    aux_out = torch.empty_like(y) if epi in (C.EPI_BIAS_QUICKGELU, C.EPI_BIAS_QUICKGELU_DERIV) else None
    colsum = torch.empty(N, dtype=torch.float32, device=DEV) if epi in (C.EPI_QUICKGELU_BWD, C.EPI_MUL_AUX_COLSUM) else None
    ws = C.workspace('linear_tn', M, N, DEV) if epi in (C.EPI_QUICKGELU_BWD, C.EPI_MUL_AUX_COLSUM) else None
    C.check(C.lib().lvl_linear_tn(C.ptr(x), C.ptr(w), C.ptr(bias), C.ptr(y), C.ptr(aux_out), C.ptr(aux_in), C.ptr(colsum),
                                  C.ptr(ws), C.ptr(sched), M, N, K, epi, C.LVL_BF16, C.stream_ptr()), 'lvl_linear_tn')
    return y, aux_out, colsum


@pytest.mark.parametrize('M,N,K', [(70001, 768, 768), (9000, 2304, 768), (5000, 768, 3072), (200, 256, 320), (3000, 512, 256)])
def test_dynamic_tile_schedule_equals_static_and_resets_its_counters(M, N, K):
    """include/lavila_hip.h, lvl_linear_tn `sched`: with a tile-counter block the persistent workgroups take their
    tiles from per-XCD device counters; every tile's arithmetic is unchanged (bit-equal outputs for all six
    epilogues, column sums included), the block is zero again when the launch has drained (so the next launch can
    reuse it), and K < 5 blocks of 64 silently keeps the static ranges."""
    from lavila_amd import _cabi as C
    g = torch.Generator(device=DEV).manual_seed(M)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device=DEV, generator=g)
    u = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    sched = torch.zeros(16, dtype=torch.int32, device=DEV)
    for epi, bias, aux in ((C.EPI_BIAS, b, None), (C.EPI_BIAS, None, None), (C.EPI_BIAS_QUICKGELU, b, None),
                           (C.EPI_QUICKGELU_BWD, None, u), (C.EPI_BIAS_RESIDUAL, b, u), (C.EPI_BIAS_QUICKGELU_DERIV, b, None),
                           (C.EPI_MUL_AUX_COLSUM, None, u)):
        want = _tn_call(x, w, bias, epi, None, aux)
        for rep in range(3):                       # the same block, back to back: it must come back zeroed
            got = _tn_call(x, w, bias, epi, sched, aux)
            for a, c in zip(got, want):
                assert (a is None) == (c is None)
                if a is not None:
                    assert torch.equal(a, c), (epi, rep)
        torch.cuda.synchronize()
        assert int(sched.abs().sum()) == 0, sched.tolist()
    # small-integer operands through the dynamic schedule: exact against the CPU
    gi = torch.Generator().manual_seed(K)
    xi = torch.randint(-2, 3, (M, K), generator=gi).float()
    wi = torch.randint(-1, 2, (N, K), generator=gi).float() * (torch.rand(N, K, generator=gi) < 0.08)
    bi = torch.randint(-8, 9, (N,), generator=gi).float()
    y, _, _ = _tn_call(xi.to(DEV).bfloat16(), wi.to(DEV).bfloat16(), bi.to(DEV), C.EPI_BIAS, sched)
    assert torch.equal(y.float().cpu(), xi @ wi.t() + bi)


def _wgrad_call(dy, x, want_dbias, sched):
    from lavila_amd import _cabi as C
    M, N = dy.shape
    K = x.shape[1]
    ws = torch.empty(int(C.lib().lvl_workspace_floats(b'linear_wgrad', N, K)), dtype=torch.float32, device=DEV)
    dw = torch.empty(N, K, dtype=torch.float32, device=DEV)
    db = torch.empty(N, dtype=torch.float32, device=DEV) if want_dbias else None
    C.check(C.lib().lvl_linear_wgrad(C.ptr(dy), C.ptr(x), C.ptr(dw), C.ptr(db), C.ptr(ws), C.ptr(sched), M, N, K,
                                     C.LVL_BF16, C.stream_ptr()), 'lvl_linear_wgrad')
    return dw, db


@pytest.mark.parametrize('M,N,K', [(200960, 768, 768), (50000, 2304, 768), (30011, 768, 3072), (8192, 1536, 512), (2000, 1024, 1024)])
def test_wgrad_chunk_schedule_equals_static_and_resets_its_counters(M, N, K):
    """lvl_linear_wgrad `sched`: row chunks handed out by device counters. On an idle GPU nobody steals: every slab
    holds the rows of the static plan in the same order, so dW (and dbias) are BIT-identical to the static schedule;
    the counter block comes back zeroed."""
    g = torch.Generator(device=DEV).manual_seed(N + K)
    dy = torch.randn(M, N, device=DEV, generator=g).bfloat16()
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    sched = torch.zeros(1024, dtype=torch.int32, device=DEV)
    for want_db in (False, True):          # (with dbias the kernel keeps the static plan and leaves the block alone)
        want = _wgrad_call(dy, x, want_db, None)
        for rep in range(2):
            got = _wgrad_call(dy, x, want_db, sched)
            assert torch.equal(got[0], want[0]), rep
            if want_db:
                assert torch.equal(got[1], want[1])
        torch.cuda.synchronize()
        assert int(sched.abs().sum()) == 0


@pytest.mark.parametrize('mod', [3, 5, 7])
def test_take_over_paths_with_late_workgroups(mod):
    """lvl_debug_late_workgroups: every mod-th workgroup of the persistent GEMMs acts as if its compute unit had been
    held by another kernel (an RCCL channel) for the whole launch. The tile queue (lvl_linear_tn) and the chunk stealing
    among the splits of a tile (lvl_linear_wgrad) must still produce every output: exact on small-integer operands,
    within float32 summation-order noise of the static schedule on random data, counters zeroed.
    (mod must not divide 8: the hook makes workgroups NEVER work, and the tile queues are per XCD = blockIdx % 8 -- a
    whole XCD that never runs has nobody to serve its queue, whereas a really late workgroup serves it when it starts.)"""
    from lavila_amd import _cabi as C
    lib = C.lib()
    assert lib.lvl_debug_late_workgroups(1) != 0
    g = torch.Generator().manual_seed(mod)
    M, N, K = 60000, 768, 768
    xi = torch.randint(-2, 3, (M, K), generator=g).float()
    wi = torch.randint(-1, 2, (N, K), generator=g).float() * (torch.rand(N, K, generator=g) < 0.08)
    bi = torch.randint(-8, 9, (N,), generator=g).float()
    dyi = torch.randint(-2, 3, (M, N), generator=g).float() * (torch.rand(M, N, generator=g) < 0.05)
    x, w, b, dy = xi.to(DEV).bfloat16(), wi.to(DEV).bfloat16(), bi.to(DEV), dyi.to(DEV).bfloat16()
    gr = torch.Generator(device=DEV).manual_seed(mod)
    xr = torch.randn(M, K, device=DEV, generator=gr).bfloat16()
    dyr = torch.randn(M, N, device=DEV, generator=gr).bfloat16()
    want_r, wantb_r = _wgrad_call(dyr, xr, True, None)
    s16 = torch.zeros(16, dtype=torch.int32, device=DEV)
    s1k = torch.zeros(1024, dtype=torch.int32, device=DEV)
    assert lib.lvl_debug_late_workgroups(mod) == 0
    try:
        for rep in range(2):
            y, _, _ = _tn_call(x, w, b, C.EPI_BIAS, s16)
            assert torch.equal(y.float().cpu(), xi @ wi.t() + bi)
            dw, db = _wgrad_call(dy, x, True, s1k)
            assert torch.equal(dw.cpu(), dyi.t() @ xi) and torch.equal(db.cpu(), dyi.sum(0))
            dwr, dbr = _wgrad_call(dyr, xr, True, s1k)
            torch.testing.assert_close(dwr, want_r, atol=2e-2, rtol=1e-4)      # |dW| ~ sqrt(M) = 245: 1e-4 relative
            torch.testing.assert_close(dbr, wantb_r, atol=2e-2, rtol=1e-4)
        torch.cuda.synchronize()
        assert int(s16.abs().sum()) == 0 and int(s1k.abs().sum()) == 0
    finally:
        assert lib.lvl_debug_late_workgroups(0) == 0


def test_dynamic_tile_schedule_on_concurrent_streams(monkeypatch):
    """The two towers launch lvl_linear_tn on two streams at once (models.py: text tower on the side stream): each
    stream draws its counter blocks from its own pool (ops.sched_block), results stay exact while the launches overlap."""
    from lavila_amd import _cabi as C
    from lavila_amd import ops
    monkeypatch.setattr(ops, 'DYNAMIC_TILES', True)       # (default: only inside a multi-rank process group)
    assert ops.sched_block(torch.device(DEV, 0)) is not None
    g = torch.Generator().manual_seed(7)
    M, N, K = 40000, 768, 768
    xi = torch.randint(-2, 3, (M, K), generator=g).float()
    wi = torch.randint(-1, 2, (N, K), generator=g).float() * (torch.rand(N, K, generator=g) < 0.08)
    want = (xi @ wi.t()).to(DEV)
    x, w = xi.to(DEV).bfloat16(), wi.to(DEV).bfloat16()
    x2, w2 = x[:8192].contiguous(), w[:512].contiguous()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    outs, outs2 = [], []
    for _ in range(6):
        outs.append(ops.linear_tn_raw(x, w, None, C.EPI_BIAS))
        with torch.cuda.stream(side):
            outs2.append(ops.linear_tn_raw(x2, w2, None, C.EPI_BIAS))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for y in outs:
        assert torch.equal(y.float(), want)
    for y in outs2:
        assert torch.equal(y.float(), want[:8192, :512])
    pools = [p for key, p in ops._sched_pools.items() if key[2] == 16]
    assert len(pools) >= 2 and all(int(p[0].abs().sum()) == 0 for p in pools)


# --------------------------------------------------------------------------------------------------------------------
# contrastive slabs at the global batch of BASELINE.json configs[2] (G = 2048 = 8 ranks x 256)
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_clip_loss_slabs_at_global_batch_2048(dtype):
    """The last rank's slabs (rows 1792..2047) of a G=2048, E=256 problem: statistics within 1e-3 of the oracle on the
    same (rounded) embeddings, argmax indices bit-exact, and the loss of the full ring of 8 slabs equals the oracle's."""
    from lavila_amd import ops
    G, B, E, row0 = 2048, 256, 256, 1792
    g = torch.Generator().manual_seed(21)
    img = O.l2_normalize(torch.randn(G, E, generator=g)).to(dtype)
    txt = O.l2_normalize(torch.randn(G, E, generator=g)).to(dtype)
    # make the task learnable-looking: pairs correlate, so the diagonal competes with 2047 negatives
    txt = O.l2_normalize((0.6 * img.float() + 0.8 * txt.float())).to(dtype)
    scale = torch.tensor([14.285714])
    stats, argmax, _ = ops.clip_loss_fwd_raw(img.to(DEV), txt.to(DEV), scale.to(DEV), B, row0)
    want_stats, want_argmax = oracle_slab_forward(img, txt, scale, B, row0)
    torch.testing.assert_close(stats.cpu(), want_stats, atol=1e-3, rtol=1e-4)
    assert torch.equal(argmax.cpu(), want_argmax)                      # int32 indices: bit-exact
    total = 0.0
    for r in range(G // B):
        st, _, _ = ops.clip_loss_fwd_raw(img.to(DEV), txt.to(DEV), scale.to(DEV), B, r * B)
        total += (st[..., 0] - st[..., 1]).sum().item()                # sum of (lse - diagonal logit) = CE sums
    loss = total / (2 * G)
    ref = O.clip_loss(img.float(), txt.float(), scale.reshape(()))['loss'].item()
    assert abs(loss - ref) < 1e-3, (loss, ref)


# --------------------------------------------------------------------------------------------------------------------
# the benchmarked model, bf16 autocast, one training step vs the float32 oracle
# --------------------------------------------------------------------------------------------------------------------
_oracle_step_cache = {}


@pytest.mark.parametrize('residual_f32', [False, True])
def test_tsfb_bf16_training_step_vs_oracle_f32(residual_f32, monkeypatch):
    """CLIP_OPENAI_TIMESFORMER_BASE (12 x 768 video blocks, 12 x 512 text blocks), 4 frames of 224^2, batch 4,
    bf16 autocast, forward + CLIPLoss + backward on the MFMA kernels, against oracle.clip_forward / clip_loss in
    float32 on the same float32 master weights.

    Error model. bf16 keeps 8 significand bits: rounding to nearest has relative error <= u = 2^-9, RMS eps = u/sqrt(3)
    = 1.1e-3. Per video block the token stream is rounded about r = 10 times on its way through (three LayerNorm
    outputs, qkv x2, attention outputs x2, projection outputs x2, the stored residual sums, the MLP hidden pair), each
    an independent relative perturbation of a branch that is O(1) of the stream. Over L = 12 blocks they add as a
    random walk: relative error of the final features ~ eps * sqrt(r * L) = 1.1e-3 * 11 = 1.2e-2; LayerNorm / softmax
    Lipschitz factors are O(1-2) here, so the unit-norm embeddings are expected within ~1.2-2.5e-2 (relative L2).
    Bound used: 2.5e-2 (measured 1.0e-2; 0.6e-2 with the float32 residual stream). A logit is 14.3 * <img, txt> with |<img, txt>| <= c: its error is at most
    14.3 * (e_img + e_txt) * max(c, e) -- here c ~ 0.1 (random towers), i.e. <= 0.07 (bound 0.1, measured 0.015); the
    loss averages 2B of them (bound 2e-2, measured 5e-6). Gradients cross every block a second time in backward with
    the same number of roundings, relative error ~ eps * sqrt(2 r L) * (1-2) = 1.7-3.4e-2 per parameter; bound used
    for EVERY parameter gradient: relative L2 <= 1e-1, with the aggregate (all parameters concatenated) <= 5e-2
    (measured: aggregate 3.7e-2, median 3.6e-2, worst tensor 5.2e-2 -- the predicted range; with the float32
    residual stream 2.0e-2 / 2.0e-2 / 3.7e-2).
    Parameters whose true gradient is ~0 (key biases: softmax is invariant to them; masked-out positional rows) are
    compared absolutely against the gradient scale of their tensor family.
    Index outputs (labels, argmax of the logits) are compared exactly whenever the oracle's own top-2 logit margin
    exceeds the logit bound.

    residual_f32=True repeats the step with LAVILA_RESIDUAL_F32 semantics (the token stream and its gradient stay
    float32 between the blocks, as under the reference's AMP): the same bounds hold, with fewer roundings per block."""
    import contextlib
    import io
    from lavila.models import models
    from lavila.models.loss import CLIPLoss
    from lavila_amd import ops
    monkeypatch.setattr(ops, 'RESIDUAL_F32', residual_f32)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = models.CLIP_OPENAI_TIMESFORMER_BASE(num_frames=4, project_embed_dim=256)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = O.procedural_weights(shapes, seed=17)
    model.load_state_dict(w)
    model.to(DEV).train()
    B = 4
    video, tokens = O.synthetic_batch(B, 4, 224, seed=31)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        out = model(video.to(DEV), tokens.to(DEV), norm_embed=True)
        crit = CLIPLoss()
        ld = crit(out)
    ld['loss'].backward()
    dbg = crit.debug_slabs(out)
    torch.cuda.synchronize()

    if 'oracle' not in _oracle_step_cache:         # the float32 CPU step is the slow part: once for both variants
        torch.set_num_threads(min(32, torch.get_num_threads() or 1))
        wo = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in w.items()}
        oo = O.clip_forward(video, tokens, wo, 12, 8, norm_embed=True)
        lo = O.clip_loss(oo['image_embed'], oo['text_embed'], oo['logit_scale'])
        lo['loss'].backward()
        _oracle_step_cache['oracle'] = (wo, oo, lo)
    wo, oo, lo = _oracle_step_cache['oracle']

    def rel(a, b):
        return ((a.float().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()
    e_img, e_txt = rel(out['image_embed'], oo['image_embed'].detach()), rel(out['text_embed'], oo['text_embed'].detach())
    dlogit = (dbg['logits'][0].cpu() - lo['logits_per_image'].detach()).abs().max().item()
    dloss = abs(ld['loss'].item() - lo['loss'].item())
    assert e_img < 2.5e-2 and e_txt < 2.5e-2, (e_img, e_txt)
    assert dlogit < 0.1 and dloss < 2e-2, (dlogit, dloss)
    assert torch.equal(dbg['labels'].cpu(), lo['labels'])
    top2 = lo['logits_per_image'].detach().topk(2, -1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * 0.1
    assert torch.equal(dbg['pred'][0].cpu()[safe], lo['pred'][safe])

    worst, num, den = [], 0.0, 0.0
    grads = dict(model.named_parameters())
    for k, p in wo.items():
        if not p.requires_grad:
            continue
        got, want = grads[k].grad, p.grad
        assert got is not None and torch.isfinite(got).all(), k
        d = (got.float().cpu() - want).norm().item()
        num += d * d
        den += want.norm().item() ** 2
        worst.append((d / max(want.norm().item(), 1e-30), d, want.norm().item(), k))
    agg = math.sqrt(num / den)
    scale = math.sqrt(den / len(worst))            # RMS gradient norm of a parameter tensor
    bad = [(r, d, n, k) for r, d, n, k in worst if r > 1e-1 and d > 1e-3 * scale]
    worst.sort(reverse=True)
    print(f'[bf16 TSF-B step, residual stream {"f32" if residual_f32 else "bf16"}] rel L2: image_embed {e_img:.2e} text_embed {e_txt:.2e}; max |d logit| {dlogit:.3f}; '
          f'|d loss| {dloss:.2e}; gradients: aggregate {agg:.2e}, worst {worst[0][0]:.2e} ({worst[0][3]}), '
          f'median {worst[len(worst) // 2][0]:.2e} over {len(worst)} tensors')
    assert agg < 5e-2, agg
    assert not bad, bad[:5]


@pytest.mark.parametrize('ctor,frames', [('CLIP_OPENAI_TIMESFORMER_LARGE', 4), ('CLIP_OPENAI_TIMESFORMER_LARGE_336PX', 2)])
def test_large_models_forward_bf16_vs_oracle(ctor, frames):
    """TSF-L/14 (D=1024, 24 blocks, 16 heads; models.py:374-491) at 224 (257 keys per frame: 8-wave MFMA space
    kernels) and at 336 (577 keys: the 4-wave large-group MFMA kernels). Forward under bf16 autocast against the float32
    oracle; bound: eps*sqrt(10*24) = 1.7e-2 relative on the unit-norm embeddings, 4e-2 used. No shape of the named
    constructors may land on the generic attention kernels (a fallback would be logged: asserted absent)."""
    import contextlib
    import io
    import warnings
    from lavila.models import models
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = getattr(models, ctor)(num_frames=frames, project_embed_dim=256)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = O.procedural_weights(shapes, seed=23)
    model.load_state_dict(w)
    model.to(DEV).eval()
    img = model.visual.patch_embed.img_size[0]
    video, tokens = O.synthetic_batch(2, frames, img, seed=41)
    with warnings.catch_warnings(record=True) as rec, torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        warnings.simplefilter('always')
        out = model(video.to(DEV), tokens.to(DEV), norm_embed=True)
    slow = [str(r.message) for r in rec if 'generic (slow) kernels' in str(r.message)]
    assert not slow, slow
    torch.set_num_threads(min(32, torch.get_num_threads() or 1))
    with torch.no_grad():
        oo = O.clip_forward(video, tokens, w, 16, 12, norm_embed=True)
    for k in ('image_embed', 'text_embed'):
        err = ((out[k].float().cpu() - oo[k]).norm() / oo[k].norm()).item()
        assert err < 4e-2, (k, err)


def test_benched_training_step_is_a_function_of_its_inputs():
    """"the result of the benched bf16 step is not proven to be a function of its inputs". The benched model
    (CLIP_OPENAI_TIMESFORMER_BASE, 12 + 12 blocks, 4 x 224^2, bf16 autocast, two towers on two streams, fused AdamW, the
    logit-scale clamp of main_pretrain.py:527-528) at a local batch of 24 -- every kernel family and launch geometry of the
    bench line -- run twice from the same weights on three batches: losses, every gradient of the last step and every
    parameter after it must agree TO THE BIT. (Until round 6 the cls token's gradient was summed with f32 atomics; at this
    clip geometry the order of the 25 additions flipped a bf16 rounding about one step in six.)"""
    import contextlib
    import io
    from lavila.models import models
    from lavila.models.loss import CLIPLoss
    B = 24

    def run():
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            model = models.CLIP_OPENAI_TIMESFORMER_BASE(num_frames=4, project_embed_dim=256)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(O.procedural_weights(shapes, seed=17))
        model.to(DEV).train()
        crit = CLIPLoss()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, eps=1e-6, fused=True)
        losses = []
        for it in range(3):
            video, tokens = O.synthetic_batch(B, 4, 224, seed=50 + it)
            opt.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                loss = crit(model(video.to(DEV), tokens.to(DEV), norm_embed=True))['loss']
            loss.backward()
            grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            opt.step()
            model.logit_scale.data.clamp_(0, 4.6052)
            losses.append(float(loss))
        torch.cuda.synchronize()
        return losses, grads, {n: p.detach().clone() for n, p in model.named_parameters()}

    l0, g0, p0 = run()
    l1, g1, p1 = run()
    assert l0 == l1, (l0, l1)
    assert all(v == v for v in l0)
    bad = [n for n in g0 if not torch.equal(g0[n], g1[n])] + [n for n in p0 if not torch.equal(p0[n], p1[n])]
    assert not bad, bad[:8]


@pytest.mark.parametrize('spread', [False, True])
def test_tsfb_bf16_step_at_batch_32_vs_the_f32_class_kernels(spread):
    """the benched bf16 instantiation was bounded at batch 4 only (1/64 of the benched batch; the CPU oracle is
    what limits that test). The float32 configuration of the SAME kernels (f32-class mode) is pinned to the reference's
    outputs at 1e-3 on six full-size fixtures (test_gpu_f32_class.py), so it can stand in for the oracle on the GPU at a
    batch the CPU cannot do in seconds: CLIP_OPENAI_TIMESFORMER_BASE, 4 x 224^2, batch 32, forward + CLIPLoss + backward under
    bf16 autocast, once with the unit-scale inputs / weights of the batch-4 oracle test (its derived bounds apply unchanged) and
    once with the "spread" ones of the round-5 fixtures (samples that do not collapse onto one embedding, ragged captions,
    sharper attention: the network amplifies a rounding 3-4x more -- DESIGN.md section 2 "conditioning" -- measured bounds),
    against the
    float32 run of the same model -- the error model of test_tsfb_bf16_training_step_vs_oracle_f32 (derivation there) with the
    bounds written at the asserts; labels exact, argmax exact wherever the float32 top-2 margin exceeds twice the logit bound."""
    import contextlib
    import io
    from lavila.models import models
    from lavila.models.loss import CLIPLoss
    B = 32
    video, tokens = O.synthetic_batch(B, 4, 224, seed=61, spread=spread)

    def run(amp):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            model = models.CLIP_OPENAI_TIMESFORMER_BASE(num_frames=4, project_embed_dim=256)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(O.procedural_weights(shapes, seed=17, spread=spread))
        model.to(DEV).train()
        crit = CLIPLoss()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
            out = model(video.to(DEV), tokens.to(DEV), norm_embed=True)
            ld = crit(out)
        ld['loss'].backward()
        dbg = crit.debug_slabs(out)
        torch.cuda.synchronize()
        return ({k: v.detach().float() for k, v in out.items()}, float(ld['loss']), dbg,
                {n: p.grad.detach().float() for n, p in model.named_parameters() if p.grad is not None})

    o32, l32, d32, g32 = run(False)
    o16, l16, d16, g16 = run(True)

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    e_img, e_txt = rel(o16['image_embed'], o32['image_embed']), rel(o16['text_embed'], o32['text_embed'])
    dlogit = (d16['logits'][0].float() - d32['logits'][0].float()).abs().max().item()
    num = sum(((g16[n] - g32[n]).norm() ** 2).item() for n in g32)
    den = sum((g32[n].norm() ** 2).item() for n in g32)
    agg = math.sqrt(num / den)
    scale = float(o32['logit_scale'])
    cmax = d32['logits'][0].float().abs().max().item() / scale            # largest |cosine| of the batch
    # unit-scale: the derived bounds (embeddings 2.5e-2, aggregate gradient 5e-2). spread: 2x larger in_proj weights in the text
    # tower / 1.5x larger qkv weights in the video tower (4x / 2.25x sharper attention scores): measured 1.2e-2 / 3.4e-2 on the
    # embeddings and 1.3e-1 on the aggregate gradient -- the float32-class path itself is amplified the same way there (2^-17
    # per product becomes 1.7e-4 on a logit); bounds at 1.5x the measurement
    b_img, b_txt, b_grad = (2.5e-2, 5e-2, 2e-1) if spread else (2.5e-2, 2.5e-2, 5e-2)
    logit_bound = scale * (b_img + b_txt) * max(cmax, b_img + b_txt)
    top2 = d32['logits'][0].float().topk(2, -1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * logit_bound
    print(f'[bf16 vs f32-class, batch {B}, spread {spread}] rel L2: image_embed {e_img:.2e} text_embed {e_txt:.2e}; max |d logit| {dlogit:.3f} '
          f'(bound {logit_bound:.3f}, largest |cos| {cmax:.2f}); |d loss| {abs(l16 - l32):.2e}; aggregate gradient {agg:.2e}; '
          f'rows with a safe argmax margin {int(safe.sum())} of {B}')
    # a logit is scale * <img, txt>: its error is at most scale * (e_img + e_txt) * max(|cos|, e)
    assert e_img < b_img and e_txt < b_txt, (e_img, e_txt)
    assert dlogit < logit_bound and abs(l16 - l32) < 2e-2 * max(1.0, abs(l32)), (dlogit, logit_bound, l16, l32)
    assert torch.equal(d16['labels'], d32['labels'])
    assert torch.equal(d16['pred'][0][safe], d32['pred'][0][safe])
    assert agg < b_grad, agg
