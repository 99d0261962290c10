"""autograd-aware wrappers around the C-ABI kernels (liblavila_hip.so).

Each `*_raw` function is a 1:1 call of one C entry point on torch device tensors; each
`torch.autograd.Function` pairs a forward kernel with its hand-written backward kernel so that the
reference's training loop (loss.backward(), DDP hooks, AdamW) works unchanged. Parameters
(gamma/beta/bias/embeddings) are always passed to the kernels as float32.
"""
import os
import weakref
from typing import Optional

import torch

from . import _cabi as C


def _f32(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if p is None:
        return None
    p = p.detach()
    if p.dtype != torch.float32:
        p = p.float()
    return p.contiguous()




def _rows_cols(x: torch.Tensor):
    cols = x.shape[-1]
    return x.numel() // cols, cols


def autocast_dtype():
    """The activation dtype of an active autocast region, or None. The kernels compute in f32 or bf16: the reference
    drivers' fp16 autocast (`torch.cuda.amp.autocast()`, main_pretrain.py:490) means bf16 here."""
    if not torch.is_autocast_enabled():
        return None
    dt = torch.get_autocast_dtype('cuda')
    return torch.bfloat16 if dt == torch.float16 else dt


def lowp(x: torch.Tensor) -> torch.Tensor:
    """fp16 activations (model.half() + images.half(): eval_zeroshot.py:212-213,256-257,293-304) are computed in bf16;
    the model wrappers hand fp16 back to the caller."""
    return x.to(torch.bfloat16) if x.dtype == torch.float16 else x


def _act(x: torch.Tensor) -> torch.Tensor:
    """Activation entering a GEMM: f32 -> the autocast dtype when autocast is on, fp16 -> bf16."""
    if x.dtype == torch.float16:
        return x.to(torch.bfloat16)
    lp = autocast_dtype()
    return x.to(lp) if (lp is not None and x.dtype == torch.float32) else x


# --------------------------------------------------------------------------------------------------
# Linear layers: hand-written MFMA GEMMs (forward / input gradient: lvl_linear_tn, weight gradient: lvl_linear_wgrad)
# --------------------------------------------------------------------------------------------------
_warned = set()
_copies = {}          # (id(parameter), shape) -> (version, bf16 copy, transposed bf16 copy, data_ptr, generation)
_generation = 0       # bumped whenever parameter VALUES may have changed behind the version counter's back


def warn_once(key, msg):
    """One line per process when a bf16 call leaves the fast kernels (shape not tiled, generic attention, ...)."""
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn('lavila_amd: ' + msg, stacklevel=3)


def invalidate_weight_cache():
    """Forget every cached bf16 weight copy (they are re-cast on their next use).

    The cache key (tensor version counter + storage pointer) does not see writes through `param.data`
    (ZeroRedundancyOptimizer's broadcast of updated shards into `param.data`, `logit_scale.data.clamp_`-style edits,
    bucket-view updates). Two automatic triggers cover the training loop: every optimizer step (a global
    `register_optimizer_step_post_hook`, installed on import) and the start of every grad-enabled model forward
    (`training_forward_begins`). Call this by hand after any other out-of-band write to a Linear weight."""
    global _generation
    _generation += 1


def training_forward_begins():
    """Called by the model wrappers (CLIP.forward / encode_*, SpaceTimeTransformer.forward*) at the top of a forward:
    with gradients enabled the copies are re-cast once per forward -- the same one lvl_cast_transpose pass per optimizer
    step as before, now independent of HOW the optimizer wrote the parameters -- and then shared by that forward, its
    backward and activation checkpointing's recomputation (which re-enters the blocks, not the wrappers)."""
    if torch.is_grad_enabled():
        invalidate_weight_cache()
        refresh_weight_copies()


_forward_depth = __import__('threading').local()


class model_forward:
    """`with ops.model_forward():` around a model-level forward (CLIP.forward / encode_*, SpaceTimeTransformer.forward*,
    VCLM_HF.encode_image): the OUTERMOST one of a grad-enabled forward bumps the weight-copy generation; nested ones
    (CLIP.forward -> encode_image -> visual.forward, the text tower started from inside the video forward) do not, so
    every weight is cast exactly once per training forward and activation checkpointing's recomputation finds the
    copies of its own forward."""

    def __enter__(self):
        d = getattr(_forward_depth, 'd', 0)
        if d == 0:
            training_forward_begins()
        _forward_depth.d = d + 1

    def __exit__(self, *exc):
        _forward_depth.d -= 1
        return False


def _optimizer_stepped(optimizer, args, kwargs):
    invalidate_weight_cache()


try:        # every torch.optim.Optimizer (ZeroRedundancyOptimizer included) announces its step() here
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(_optimizer_stepped)
except ImportError:         # older torch: the per-forward trigger alone
    pass


# float32 activations (the parity configuration, north_star "within 1e-3 fp32"): the SAME MFMA GEMM kernels in their
# f32-class mode -- operands as bf16 term images (split3 / lvl_split_bf16x3), float32 results; ~2^-17 relative per
# product instead of bf16's 2^-9. LAVILA_F32_MFMA=0 sends float32 Linears to the library GEMM instead (A/B only).
F32_MFMA = os.environ.get('LAVILA_F32_MFMA', '1') != '0'


def split3(x2: torch.Tensor, role: int, stack: bool = False) -> torch.Tensor:
    """float32 [R, C] -> its three bf16 term images (h, h, l) (role 0: the x / dy side of a product) or (h, l, h)
    (role 1: the w / x side): side by side along the contraction, [R, 3C] (operand of lvl_linear_tn's f32-class mode),
    or -- stack=True -- one under the other, [3*Rp, C] (operands of lvl_linear_wgrad, which contracts over rows; rows
    padded with zeros so that the kernel sees at least one 32-row step)."""
    C.require_device(x2)
    R, Cc = x2.shape
    if stack:
        Rp = max(R, 11)
        out = (torch.zeros if Rp != R else torch.empty)(3 * Rp, Cc, dtype=torch.bfloat16, device=x2.device)
        rs, ts = Cc, Rp * Cc
    else:
        out = torch.empty(R, 3 * Cc, dtype=torch.bfloat16, device=x2.device)
        rs, ts = 3 * Cc, Cc
    C.check(C.lib().lvl_split_bf16x3(C.ptr(x2), C.ptr(out), R, Cc, Cc, rs, ts, role, C.stream_ptr()), 'lvl_split_bf16x3')
    return out


def _split_pair(src):
    src = src.float().contiguous()
    return split3(src, 1), split3(src.t().contiguous(), 1)


def _cast_pair(src):
    if src.dtype == torch.float32 and src.is_contiguous():
        w = torch.empty_like(src, dtype=torch.bfloat16)
        wt = torch.empty(src.shape[1], src.shape[0], dtype=torch.bfloat16, device=src.device)
        C.check(C.lib().lvl_cast_transpose(C.ptr(src), C.ptr(w), C.ptr(wt), src.shape[0], src.shape[1], C.stream_ptr()),
                'lvl_cast_transpose')
    else:                   # fp16 parameters (model.half(), eval_zeroshot.py --use-half) and views
        w = src.to(torch.bfloat16).contiguous()
        wt = w.t().contiguous()
    return w, wt


def weight_copies(weight: torch.Tensor, f32: bool = False):
    """f32=True: the f32-class operands instead -- (w3 [out, 3 in], wt3 [in, 3 out]) term images of the weight and of its
    transpose (split3 role 1) -- cached under the same rules.
    (w, wt): the bf16 copy [out,in] the forward GEMM reads and the transposed bf16 copy [in,out] the input-gradient
    GEMM reads (both operands of lvl_linear_tn are contraction-contiguous). One lvl_cast_transpose pass per optimizer
    step: the pair is cached per parameter and keyed by (version counter, storage pointer, cache generation), so
    activation checkpointing's second forward and every backward reuse it; see invalidate_weight_cache for what bumps
    the generation. Under hipGraph capture the cache is bypassed: the cast is recorded INTO the graph (into
    graph-owned buffers), so a replay after a weight update reads the updated weights."""
    src = weight.detach()
    make = _split_pair if f32 else _cast_pair
    if src.is_cuda and torch.cuda.is_current_stream_capturing():
        return make(src)
    # keyed by the identity of the parameter (a 2-D view of one -- the Conv2d weight of the patch embedding -- keys on
    # its base); a weakref finaliser drops the entry with the parameter, nothing is attached to the parameter itself
    # (pickling / deepcopy of the model see no extra state)
    holder = weight._base if weight._base is not None else weight
    ver, key = weight._version, (id(holder), tuple(weight.shape), tuple(weight.stride()), f32)
    c = _copies.get(key)
    if c is not None and c[0] == ver and c[3] == weight.data_ptr() and c[4] == _generation:
        return c[1], c[2]
    if c is None:
        weakref.finalize(holder, _copies.pop, key, None)
    w, wt = make(src)
    # (the parameter itself, weakly, when it is one -- refresh_weight_copies re-casts those entries in one launch)
    _copies[key] = (ver, w, wt, weight.data_ptr(), _generation, weakref.ref(weight) if weight._base is None else None)
    return w, wt


# LAVILA_WEIGHT_REFRESH=1: every stale weight copy is re-cast by ONE launch at the top of the training forward instead of one
# lvl_cast_transpose launch per weight at its first use. Off by default: measured neutral to slightly slower (same box,
# alternating: 163.86 ms per step lazily, 164.21 with the refresh -- the ~120 five-microsecond launches already hide between the
# step's kernels, the single 0.3-ms pass at the top does not; profiles/r06_weight_refresh.txt).
WEIGHT_REFRESH = os.environ.get('LAVILA_WEIGHT_REFRESH', '0') == '1'


def refresh_weight_copies():
    """Re-cast, in ONE lvl_cast_transpose_multi launch per device, every cached (bf16, transposed bf16) pair that the new
    cache generation made stale. Called at the top of a grad-enabled model forward (training_forward_begins), i.e. exactly
    where the first use of each weight would otherwise re-cast it: the same values at the same point of the step, 1 launch
    instead of ~120 (main_pretrain.py:520-533: optimizer.step(), then the next model(...)). Entries whose parameter is gone,
    moved, not float32-contiguous, a view (the patch embedding's reshaped Conv2d weight) or an f32-class image pair keep the
    lazy path; so does everything under hipGraph capture."""
    if not WEIGHT_REFRESH or not _copies or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
        return
    by_dev = {}
    for key, c in list(_copies.items()):
        if key[3] or len(c) < 6 or c[5] is None or c[4] < _generation - 3:      # (not used by the last step: lazily)
            continue
        w = c[5]()
        if (w is None or not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or not w.is_contiguous()
                or w.data_ptr() != c[3]):
            continue
        by_dev.setdefault(w.device, []).append((key, w))
    for dev, items in by_dev.items():
        if len(items) < 2 or dev != torch.device('cuda', torch.cuda.current_device()):
            continue
        rows, outs, tile0 = [], [], 0
        for key, w in items:
            src = w.detach()
            n, k = src.shape
            a = torch.empty_like(src, dtype=torch.bfloat16)
            b = torch.empty(k, n, dtype=torch.bfloat16, device=dev)
            rows.append([src.data_ptr(), a.data_ptr(), b.data_ptr(), n | (k << 32), tile0])
            tile0 += ((n + 63) // 64) * ((k + 63) // 64)
            outs.append((key, w, a, b))
        table = torch.tensor(rows, dtype=torch.int64, device=dev)
        C.check(C.lib().lvl_cast_transpose_multi(C.ptr(table), len(rows), tile0, C.stream_ptr()), 'lvl_cast_transpose_multi')
        for key, w, a, b in outs:
            _copies[key] = (w._version, a, b, w.data_ptr(), _generation, weakref.ref(w))


def _tn_ok(rows: int, n_out: int, n_in: int) -> bool:
    """lvl_linear_tn tiles N in 256s and K in 64s (every width of the CLIP_OPENAI_TIMESFORMER_* towers)."""
    return rows > 0 and n_out % 256 == 0 and n_in % 64 == 0 and rows * max(n_in, n_out) * 2 < (1 << 32)


def _wgrad(dy2, x2, wdt):
    """dW = dy^T x: the MFMA weight-gradient kernel; shapes it does not tile fall back to a library GEMM (logged)."""
    rows, n_out, n_in = dy2.shape[0], dy2.shape[1], x2.shape[1]
    ws_floats = -1
    if (rows > 0 and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and dy2.is_contiguous()
            and x2.is_contiguous() and dy2.is_cuda):
        ws_floats = C.lib().lvl_workspace_floats(b'linear_wgrad', n_out, n_in)
    if ws_floats >= 0:
        if rows < 32:       # the class-token / EOT-row Linears of a small batch: zero rows up to the kernel's one 32-row step
            dy2 = torch.cat([dy2, dy2.new_zeros(32 - rows, n_out)])
            x2 = torch.cat([x2, x2.new_zeros(32 - rows, n_in)])
        return linear_wgrad_raw(dy2, x2, False, int(ws_floats))[0].to(wdt)
    if dy2.dtype == torch.bfloat16 and rows >= 4096:
        warn_once(('wgrad', n_out, n_in), f'weight gradient [{n_out},{n_in}] falls back to a library GEMM '
                                          '(lvl_linear_wgrad tiles multiples of 192/288/384 or 128/256)')
    return (dy2.t() @ x2).to(wdt)


def _wgrad_f32(dy2, x2, wdt):
    """dW = dy^T x for float32 operands: lvl_linear_wgrad on row-stacked term images (f32-class, see split3)."""
    n_out, n_in = dy2.shape[1], x2.shape[1]
    ws_floats = C.lib().lvl_workspace_floats(b'linear_wgrad', n_out, n_in) if (F32_MFMA and dy2.is_cuda) else -1
    if ws_floats < 0 or dy2.shape[0] * max(n_in, n_out) * 6 >= (1 << 32):
        return (dy2.t() @ x2).to(wdt)
    dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    return linear_wgrad_raw(split3(dy2, 0, stack=True), split3(x2, 1, stack=True), False, int(ws_floats))[0].to(wdt)


def _tn_ok_f32(rows: int, n_out: int, n_in: int) -> bool:
    return F32_MFMA and _tn_ok(rows, n_out, 3 * n_in)


class _LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) for token-major activations [rows, in].

    bf16: forward and input gradient are lvl_linear_tn calls (the input gradient multiplies by the cached transposed
    weight copy, so both GEMMs read contraction-contiguous operands), the weight gradient is lvl_linear_wgrad. f32
    (the parity configuration): the same three kernels in f32-class mode (bf16 term images in, float32 out: split3).
    Widths the kernels do not tile use the library GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        rows, n_in, n_out = x2.shape[0], x2.shape[1], weight.shape[0]
        own = (x.dtype == torch.bfloat16 and x.is_cuda and _tn_ok(rows, n_out, n_in)
               and (not ctx.needs_input_grad[0] or _tn_ok(rows, n_in, n_out)))     # the input gradient swaps N and K
        ctx.own = own
        ctx.meta = (weight.dtype, None if bias is None else bias.dtype, x.shape)
        ctx.f32own = (not own and x.dtype == torch.float32 and x.is_cuda and _tn_ok_f32(rows, n_out, n_in)
                      and (not ctx.needs_input_grad[0] or _tn_ok_f32(rows, n_in, n_out)))
        if ctx.f32own:          # f32-class mode of the same kernels (the parity configuration)
            w3, wt3 = weight_copies(weight, f32=True)
            x2 = x2 if x2.is_contiguous() else x2.contiguous()
            ctx.save_for_backward(x2, wt3)
            y = linear_tn_raw(split3(x2, 0), w3, _f32(bias), C.EPI_BIAS, f32=True)
            return y.reshape(*x.shape[:-1], n_out)
        if own:
            w, wt = weight_copies(weight)
            ctx.save_for_backward(x2 if x2.is_contiguous() else x2.contiguous(), wt)
            y = linear_tn_raw(x2 if x2.is_contiguous() else x2.contiguous(), w, _f32(bias), C.EPI_BIAS)
            return y.reshape(*x.shape[:-1], n_out)
        if x.dtype == torch.bfloat16 and rows >= 4096:
            warn_once(('linear', n_out, n_in), f'Linear [{n_out},{n_in}] runs on the library GEMM (lvl_linear_tn needs '
                                               'out % 256 == 0 and in % 64 == 0)')
        w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
        b = None if bias is None else (bias if bias.dtype == x.dtype else bias.to(x.dtype))
        ctx.save_for_backward(x2, w)
        with torch.autocast('cuda', enabled=False):
            return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors                 # own path: w is the TRANSPOSED copy [in, out]
        wdt, bdt, xshape = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = dw = db = None
        with torch.autocast('cuda', enabled=False):
            if ctx.f32own:
                dy2 = dy2.float()
                if ctx.needs_input_grad[0]:
                    dx = linear_tn_raw(split3(dy2, 0), w, None, C.EPI_BIAS, f32=True).reshape(xshape)
                if ctx.needs_input_grad[1]:
                    dw = _wgrad_f32(dy2, x2, wdt)
                if bdt is not None and ctx.needs_input_grad[2]:
                    db = dy2.sum(0).to(bdt)
                return dx, dw, db
            if ctx.needs_input_grad[0]:
                dx = (linear_tn_raw(dy2, w, None, C.EPI_BIAS) if ctx.own else dy2 @ w).reshape(xshape)
            if ctx.needs_input_grad[1]:
                dw = _wgrad(dy2, x2, wdt) if (ctx.own or x2.dtype != torch.float32) else _wgrad_f32(dy2, x2, wdt)
            if bdt is not None and ctx.needs_input_grad[2]:
                db = dy2.sum(0, dtype=torch.float32).to(bdt)      # f32 accumulation AND f32 result
        return dx, dw, db


# What the fused MLP keeps for its backward in bf16: quickgelu'(u) (default) or the pre-activation u (LAVILA_GELU_DERIV=0).
GELU_DERIV = os.environ.get('LAVILA_GELU_DERIV', '1') != '0'


class _MlpFn(torch.autograd.Function):
    """y = fc2(QuickGELU(fc1(x) + b1)) without fc2's bias (the caller leaves it pending for the next fused
    residual + LayerNorm): timesformer.py:52-58 / openai_model.py:189-192. Two lvl_linear_tn calls forward (the
    first one adds the bias, applies the activation and keeps what the backward needs), and in backward the QuickGELU
    derivative and the fc1 bias gradient are the epilogue of fc2's input-gradient GEMM: the [rows, 4D] tensors are
    touched by GEMM epilogues only. What is kept (GELU_DERIV): bf16 keeps quickgelu'(u) itself -- the forward epilogue
    has sigmoid(1.702 u) in a register, the backward epilogue then is one multiply (LVL_EPI_BIAS_QUICKGELU_DERIV /
    LVL_EPI_MUL_AUX_COLSUM; LAVILA_GELU_DERIV=0 keeps the pre-activation u as the f32-class mode always does)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        f32 = ctx.f32 = x.dtype == torch.float32        # f32-class mode: term images in, float32 u / a / y
        w1b, w1t = weight_copies(w1, f32)
        w2b, w2t = weight_copies(w2, f32)
        ctx.deriv = GELU_DERIV and not f32
        a, u = linear_tn_raw(split3(x2, 0) if f32 else x2, w1b, _f32(b1),
                             C.EPI_BIAS_QUICKGELU_DERIV if ctx.deriv else C.EPI_BIAS_QUICKGELU, f32=f32)
        y = linear_tn_raw(split3(a, 0) if f32 else a, w2b, None, C.EPI_BIAS, f32=f32)
        ctx.save_for_backward(x2, u, a, w1t, w2t)
        ctx.meta = (w1.dtype, None if b1 is None else b1.dtype, w2.dtype, x.shape)
        return y.reshape(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, u, a, w1t, w2t = ctx.saved_tensors
        w1dt, b1dt, w2dt, xshape = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if ctx.f32:
            dy2 = dy2.float()
            du, db1 = linear_tn_raw(split3(dy2, 0), w2t, None, C.EPI_QUICKGELU_BWD, aux_in=u, f32=True)
            dw2 = _wgrad_f32(dy2, a, w2dt) if ctx.needs_input_grad[3] else None
            dx = (linear_tn_raw(split3(du, 0), w1t, None, C.EPI_BIAS, f32=True).reshape(xshape)
                  if ctx.needs_input_grad[0] else None)
            dw1 = _wgrad_f32(du, x2, w1dt) if ctx.needs_input_grad[1] else None
            return dx, dw1, (db1.to(b1dt) if (b1dt is not None and ctx.needs_input_grad[2]) else None), dw2
        du, db1 = linear_tn_raw(dy2, w2t, None, C.EPI_MUL_AUX_COLSUM if ctx.deriv else C.EPI_QUICKGELU_BWD, aux_in=u)
        dw2 = _wgrad(dy2, a, w2dt) if ctx.needs_input_grad[3] else None
        dx = linear_tn_raw(du, w1t, None, C.EPI_BIAS).reshape(xshape) if ctx.needs_input_grad[0] else None
        dw1 = _wgrad(du, x2, w1dt) if ctx.needs_input_grad[1] else None
        return dx, dw1, (db1.to(b1dt) if (b1dt is not None and ctx.needs_input_grad[2]) else None), dw2


def mlp_quickgelu(x, w1, b1, w2):
    """fc2(QuickGELU(fc1(x))) minus fc2's bias; fused GEMM epilogues in bf16, composed kernels otherwise."""
    x = _act(x)
    rows = x.numel() // x.shape[-1]
    if (x.dtype == torch.bfloat16 and x.is_cuda and b1 is not None and _tn_ok(rows, w1.shape[0], w1.shape[1])
            and _tn_ok(rows, w1.shape[1], w1.shape[0]) and _tn_ok(rows, w2.shape[0], w2.shape[1])
            and _tn_ok(rows, w2.shape[1], w2.shape[0])):
        return _MlpFn.apply(x, w1, b1, w2)
    if (x.dtype == torch.float32 and x.is_cuda and b1 is not None and w1.dtype == torch.float32
            and _tn_ok_f32(rows, w1.shape[0], w1.shape[1]) and _tn_ok_f32(rows, w1.shape[1], w1.shape[0])
            and _tn_ok_f32(rows, w2.shape[0], w2.shape[1]) and _tn_ok_f32(rows, w2.shape[1], w2.shape[0])):
        return _MlpFn.apply(x, w1, b1, w2)          # the same fused epilogues in f32-class mode
    return linear(bias_quick_gelu(linear(x, w1), b1), w2)


def linear_wgrad_raw(dy, x, want_dbias: bool, ws_floats: int = -1):
    """dW [N,K] f32 = dy[M,N]^T x[M,K] (+ dbias [N] f32) through lvl_linear_wgrad (bf16 operands)."""
    C.require_device(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    if ws_floats < 0:
        ws_floats = C.lib().lvl_workspace_floats(b'linear_wgrad', N, K)
        if ws_floats < 0:
            raise C.HipExtensionError(f'lvl_linear_wgrad: no tiling for N={N} K={K}')
    ws = torch.empty(int(ws_floats), dtype=torch.float32, device=dy.device)
    dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_dbias else None
    C.check(C.lib().lvl_linear_wgrad(C.ptr(dy), C.ptr(x), C.ptr(dw), C.ptr(db), C.ptr(ws),
                                     C.ptr(sched_block(dy.device, 1024)), M, N, K, C.dtype_code(dy), C.stream_ptr()),
            'lvl_linear_wgrad')
    return dw, db


# Tile-counter blocks of the persistent GEMM (lvl_linear_tn `sched`): 16 x uint32, zero on entry, zeroed again by the
# kernel's last workgroup. One zero-filled pool per (device, stream), handed out round-robin: launches of one stream
# run in order, so a block is long back to zero when its turn comes again (4096 launches later), and the two towers'
# streams never share a block. Under hipGraph capture the block is allocated (and zeroed) inside the graph.
# lvl_linear_wgrad's chunk-counter blocks (1024 words) come from a second pool of the same kind.
_SCHED_BLOCKS = {16: 4096, 1024: 512}
_sched_pools = {}
# LAVILA_DYNAMIC_TILES: '1' always, '0' never, unset = when this process is one rank of a multi-GPU job (an initialised
# torch.distributed group of more than one rank: RCCL's channel kernels then share the GPU with the step -- that is
# the situation the counters are for; measured with 16 CUs held by another kernel: -5..8 % instead of -26..30 %,
# profiles/r03_cu_contention.txt). A single-GPU run keeps the static ranges: the counters cost 0.8 % of the step
# (the one synchronous hand-out per launch and the claims' round trips).
DYNAMIC_TILES = {'1': True, '0': False}.get(os.environ.get('LAVILA_DYNAMIC_TILES', ''), None)


def dynamic_tiles() -> bool:
    if DYNAMIC_TILES is not None:
        return DYNAMIC_TILES
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def sched_block(device, words=16):
    if not dynamic_tiles():
        return None
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(words, dtype=torch.int32, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream, words)
    pool = _sched_pools.get(key)
    if pool is None:
        pool = _sched_pools[key] = [torch.zeros(_SCHED_BLOCKS[words], words, dtype=torch.int32, device=device), 0]
    pool[1] = (pool[1] + 1) % _SCHED_BLOCKS[words]
    return pool[0][pool[1]]


def linear_tn_raw(x, w, bias=None, epilogue=C.EPI_BIAS, aux_in=None, f32=False):
    """One lvl_linear_tn call on bf16 tensors: y[M,N] = epilogue(x[M,K] . w[N,K]^T).
    Returns y (EPI_BIAS; EPI_BIAS_RESIDUAL: + aux_in [M,N]), (y, u) (EPI_BIAS_QUICKGELU), (y, quickgelu'(u))
    (EPI_BIAS_QUICKGELU_DERIV) or (y, colsum) (EPI_QUICKGELU_BWD, EPI_MUL_AUX_COLSUM).
    f32=True: the kernel's f32-class mode -- x, w are bf16 term images (split3), y / u / aux_in float32."""
    C.require_device(x, w, bias, aux_in)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=x.device)
    aux_out = colsum = ws = None
    if epilogue in (C.EPI_BIAS_QUICKGELU, C.EPI_BIAS_QUICKGELU_DERIV):
        aux_out = torch.empty_like(y)
    elif epilogue in (C.EPI_QUICKGELU_BWD, C.EPI_MUL_AUX_COLSUM):
        colsum = torch.empty(N, dtype=torch.float32, device=x.device)
        ws = C.workspace('linear_tn', M, N, x.device)
    C.check(C.lib().lvl_linear_tn(C.ptr(x), C.ptr(w), C.ptr(bias), C.ptr(y), C.ptr(aux_out), C.ptr(aux_in),
                                  C.ptr(colsum), C.ptr(ws), C.ptr(sched_block(x.device)), M, N, K, epilogue,
                                  C.LVL_F32 if f32 else C.LVL_BF16, C.stream_ptr()), 'lvl_linear_tn')
    if epilogue in (C.EPI_BIAS_QUICKGELU, C.EPI_BIAS_QUICKGELU_DERIV):
        return y, aux_out
    if epilogue in (C.EPI_QUICKGELU_BWD, C.EPI_MUL_AUX_COLSUM):
        return y, colsum
    return y


def linear_skinny_raw(x, w, bias=None, act=None):
    """One lvl_linear_skinny call on bf16 tensors: y[M,N] = act(x[M,K] . w[N,K]^T + bias) -- few rows: strips, many rows
    or 50432 columns: LDS-staged tiles (gemm_skinny.hip). act: None, C.ACT_GELU_NEW or C.ACT_SQRELU."""
    C.require_device(x, w, bias)
    M, K = x.shape
    y = torch.empty(M, w.shape[0], dtype=torch.bfloat16, device=x.device)
    C.check(C.lib().lvl_linear_skinny(C.ptr(x), C.ptr(w), C.ptr(bias), C.ptr(y), M, w.shape[0], K,
                                      -1 if act is None else act, C.stream_ptr()), 'lvl_linear_skinny')
    return y


def linear_skinny_f32c_raw(x3, w3, bias=None, act=None):
    """One lvl_linear_skinny_f32c call: y[M,N] float32 = act(x . w^T + bias) for float32 x [M,K], w [N,K] handed over as
    their bf16 term images x3 = split3(x, 0) [M,3K], w3 = split3(w, 1) [N,3K] (f32-class mode, ~2^-17 per product)."""
    C.require_device(x3, w3, bias)
    M, K3 = x3.shape
    y = torch.empty(M, w3.shape[0], dtype=torch.float32, device=x3.device)
    C.check(C.lib().lvl_linear_skinny_f32c(C.ptr(x3), C.ptr(w3), C.ptr(bias), C.ptr(y), M, w3.shape[0], K3,
                                           -1 if act is None else act, C.stream_ptr()), 'lvl_linear_skinny_f32c')
    return y


def linear_f32_rows(x2, w3, bias=None, act=None):
    """float32 x2 [M,K] against the term images w3 [N,3K] of a float32 weight: the 256-column-panel kernel's f32-class mode
    where it tiles and there are enough rows to fill it, lvl_linear_skinny_f32c otherwise (N % 16 == 0, K % 32 == 0).
    Inference helper (no autograd): the narrator's float32 decoder and ops.linear under no_grad."""
    M, K = x2.shape
    N = w3.shape[0]
    x3 = split3(x2 if x2.is_contiguous() else x2.contiguous(), 0)
    if M > 256 and _tn_ok(M, N, 3 * K):
        y = linear_tn_raw(x3, w3, bias, C.EPI_BIAS, f32=True)
        if act is not None:
            C.check(C.lib().lvl_act_inplace(C.ptr(y), y.numel(), act, C.dtype_code(y), C.stream_ptr()), 'lvl_act_inplace')
        return y
    return linear_skinny_f32c_raw(x3, w3, bias, act)


# Column-sum tokens (round 5). The bias gradient of a qkv Linear needs sum_rows(dout) of its attention core (the v third:
# every softmax row sums to 1). dout is the INPUT gradient of the projection Linear behind the attention, dout = dy W, so
# sum_rows(dout) = sum_rows(dy) W -- and sum_rows(dy) is the projection's own bias gradient, which the LayerNorm backward
# behind it produces anyway. A "token" is a [width] float32 tensor that travels beside an activation through the own
# autograd Functions (attention core -> projection -> fused add / LayerNorm); its GRADIENT is defined as the column sums of
# the activation's gradient, each Function computing it from what it already has (a [D] x [D, D] vector-matrix product,
# lvl_vec_mat_f32, a few microseconds) instead of a 300 MB pass over dout per attention. Exact in real arithmetic; in
# floating point it is the more accurate of the two (float32 sums of float32 column sums against sums of bf16-rounded
# rows). LAVILA_COLSUM_TOKENS=0 switches the tokens off (the attention backward then reduces dout itself, as before).
COLSUM_TOKENS = os.environ.get('LAVILA_COLSUM_TOKENS', '1') != '0'


def vec_mat(v, weight):
    """v [N] float32 . weight [N, K] -> [K] float32 (lvl_vec_mat_f32)."""
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    v = v.detach().float().contiguous()
    C.require_device(v, w)
    out = torch.empty(w.shape[1], dtype=torch.float32, device=w.device)
    C.check(C.lib().lvl_vec_mat_f32(C.ptr(v), C.ptr(w), C.ptr(out), w.shape[0], w.shape[1], C.stream_ptr()), 'lvl_vec_mat_f32')
    return out


class _LinearTokenFn(torch.autograd.Function):
    """(y, ytoken) = (x W^T, token of y) for a bf16 activation x that carries the column-sum token `xtoken`: _LinearFn's
    own-kernel path without a bias (the caller leaves it pending for the fused add + LayerNorm), plus the token rule
    d(xtoken) = d(ytoken) . W."""

    @staticmethod
    def forward(ctx, x, weight, xtoken):
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w, wt = weight_copies(weight)
        ctx.save_for_backward(x2, wt, weight)
        ctx.meta = (weight.dtype, x.shape)
        ctx.set_materialize_grads(False)        # an unused token arrives as None, not as zeros
        y = linear_tn_raw(x2, w, None, C.EPI_BIAS)
        return y.reshape(*x.shape[:-1], weight.shape[0]), x.new_zeros(weight.shape[0], dtype=torch.float32)   # token VALUE: unused, defined (zeros)

    @staticmethod
    def backward(ctx, dy, dytoken):
        x2, wt, weight = ctx.saved_tensors
        wdt, xshape = ctx.meta
        if dy is None:
            return None, None, None
        dy2 = dy.reshape(-1, dy.shape[-1])
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        with torch.autocast('cuda', enabled=False):
            dx = linear_tn_raw(dy2, wt, None, C.EPI_BIAS).reshape(xshape) if ctx.needs_input_grad[0] else None
            dw = _wgrad(dy2, x2, wdt) if ctx.needs_input_grad[1] else None
            dtok = vec_mat(dytoken, weight) if (dytoken is not None and ctx.needs_input_grad[2]) else None
        return dx, dw, dtok


def linear_with_token(x, weight, xtoken):
    """(y, ytoken) of a bias-free Linear on a token-carrying bf16 activation, or (linear(x, weight), None) when the shapes
    are not the own kernels' (the token chain then simply ends: consumers fall back to reducing their rows)."""
    x = _act(x)
    rows, n_in, n_out = x.numel() // x.shape[-1], x.shape[-1], weight.shape[0]
    if (xtoken is not None and x.dtype == torch.bfloat16 and x.is_cuda and weight.dtype == torch.float32
            and _tn_ok(rows, n_out, n_in) and _tn_ok(rows, n_in, n_out)):
        return _LinearTokenFn.apply(x, weight, xtoken)
    return linear(x, weight), None


def project(x, proj):
    """x @ proj for the [width, embed_dim] projection parameters (models.py:146,161): the Linear kernels against the
    transposed view (its gradient flows back to `proj` through the view)."""
    return linear(x, proj.t())


def linear(x, weight, bias=None):
    """nn.Linear forward with hand-written forward / input-gradient / weight-gradient GEMMs behind it.
    Activation dtype = x.dtype (the autocast dtype when autocast is on; fp16 -> bf16); parameters stay masters."""
    x = _act(x)
    if x.dtype == torch.bfloat16 and x.is_cuda and not (torch.is_grad_enabled() and (
            x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad))):
        # inference, widths the 256-column-panel kernel does not tile (the narrator's pooling projection to_kv
        # [128 x 768], coca.py:78): the strip / LDS-tile kernel of the decoder (lvl_linear_skinny) instead of the library
        rows, n_in, n_out = x.numel() // x.shape[-1], x.shape[-1], weight.shape[0]
        if (rows > 0 and not _tn_ok(rows, n_out, n_in) and n_out % 16 == 0 and n_in % 64 == 0
                and rows * max(n_in, n_out) * 2 < (1 << 31)):
            x2 = x.reshape(-1, n_in)
            return linear_skinny_raw(x2 if x2.is_contiguous() else x2.contiguous(), weight_copies(weight)[0],
                                     _f32(bias)).reshape(*x.shape[:-1], n_out)
    if F32_MFMA and x.dtype == torch.float32 and x.is_cuda and weight.dtype == torch.float32 and not (
            torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or
                                         (bias is not None and bias.requires_grad))):
        # float32 inference on widths the panel kernel does not tile (small towers, the narrator's pooling projections,
        # round 5): the strip kernel's f32-class mode instead of the library GEMM
        rows, n_in, n_out = x.numel() // x.shape[-1], x.shape[-1], weight.shape[0]
        if (rows > 0 and not _tn_ok_f32(rows, n_out, n_in) and n_out % 16 == 0 and n_in % 32 == 0
                and rows * max(3 * n_in, n_out) * 4 < (1 << 31)):
            return linear_f32_rows(x.reshape(-1, n_in), weight_copies(weight, f32=True)[0],
                                   _f32(bias)).reshape(*x.shape[:-1], n_out)
    return _LinearFn.apply(x, weight, bias)


# --------------------------------------------------------------------------------------------------
# LayerNorm (optionally fused with residual + bias add)
# --------------------------------------------------------------------------------------------------
def layernorm_fwd_raw(x, x2, xbias, gamma, beta, eps, keep_sum):
    C.require_device(x, x2, xbias, gamma, beta)
    rows, cols = _rows_cols(x)
    y = torch.empty_like(x)
    s = torch.empty_like(x) if keep_sum else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    C.check(C.lib().lvl_layernorm_fwd(C.ptr(x), C.ptr(x2), C.ptr(xbias), C.ptr(gamma), C.ptr(beta), C.ptr(s),
                                      C.ptr(y), C.ptr(mean), C.ptr(rstd), rows, cols, float(eps),
                                      C.dtype_code(x), C.stream_ptr()), 'lvl_layernorm_fwd')
    return y, s, mean, rstd


def layernorm_bwd_raw(dy, x, x2, xbias, gamma, mean, rstd, dadd, want_dxsum, want_plain=False):
    """want_plain: also return the normalisation's own input gradient without dadd (dx = dx_plain + dadd)."""
    C.require_device(dy, x, x2, xbias, gamma, mean, rstd, dadd)
    rows, cols = _rows_cols(x)
    dx = torch.empty_like(x)
    dx_plain = torch.empty_like(x) if want_plain else None
    dgamma = torch.empty(cols, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(cols, dtype=torch.float32, device=x.device)
    dxsum = torch.empty(cols, dtype=torch.float32, device=x.device) if want_dxsum else None
    ws = C.workspace('layernorm_bwd', rows, cols, x.device)
    C.check(C.lib().lvl_layernorm_bwd(C.ptr(dy), C.ptr(x), C.ptr(x2), C.ptr(xbias), C.ptr(gamma), C.ptr(mean),
                                      C.ptr(rstd), C.ptr(dadd), C.ptr(dx), C.ptr(dx_plain), C.ptr(dgamma),
                                      C.ptr(dbeta), C.ptr(dxsum), C.ptr(ws), rows, cols, C.dtype_code(x),
                                      C.stream_ptr()), 'lvl_layernorm_bwd')
    if want_plain:
        return dx, dgamma, dbeta, dxsum, dx_plain
    return dx, dgamma, dbeta, dxsum


class _LayerNormFn(torch.autograd.Function):
    """y = LN(x) ; see lvl_layernorm_fwd/bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        g, b = _f32(weight), _f32(bias)
        y, _, mean, rstd = layernorm_fwd_raw(x, None, None, g, b, eps, False)
        ctx.save_for_backward(x, g, mean, rstd)
        ctx.pdt = (weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd = ctx.saved_tensors
        dx, dg, db, _ = layernorm_bwd_raw(dy.contiguous(), x, None, None, g, mean, rstd, None, False)
        return dx, dg.to(ctx.pdt[0]), db.to(ctx.pdt[1]), None


# Residual stream dtype under autocast. Default: the autocast dtype (bf16) from the patch embedding to the final norm
# (INTEGRATION.md section 3). LAVILA_RESIDUAL_F32=1 (or ops.RESIDUAL_F32 = True) keeps the stream -- and its gradient --
# in float32 as the reference's AMP does (timesformer.py:353-366, 183-196: cat with the f32 cls_token, f32 LayerNorm
# outputs, only GEMM outputs are half): the LayerNorm kernels then run in their f32 instantiation, branch outputs are
# widened on the way in and the normalised rows narrowed on the way out (two extra element-wise passes per site: a
# fidelity mode, not a fast path).
RESIDUAL_F32 = os.environ.get('LAVILA_RESIDUAL_F32', '0') == '1'


def _gemm_input_dtype():
    """dtype a LayerNorm output takes when it feeds GEMMs: the autocast dtype, or None (= leave as is)."""
    return autocast_dtype()


def _narrow(h):
    lp = _gemm_input_dtype()
    return h.to(lp) if (lp is not None and h.dtype == torch.float32) else h


def layer_norm(x, weight, bias, eps, stream=False):
    """stream=True: the output IS the residual stream (ln_pre): it keeps the dtype of x."""
    y = _LayerNormFn.apply(lowp(x), weight, bias, eps)
    return y if stream else _narrow(y)


class _AddLayerNormFn(torch.autograd.Function):
    """(s, h) = (res + y + ybias, LN(res + y + ybias)); s is only materialised when keep_sum."""

    @staticmethod
    def forward(ctx, res, y, ybias, weight, bias, eps, keep_sum):
        res, y = res.contiguous(), y.contiguous()
        yb, g, b = _f32(ybias), _f32(weight), _f32(bias)
        h, s, mean, rstd = layernorm_fwd_raw(res, y, yb, g, b, eps, keep_sum)
        if keep_sum:
            ctx.save_for_backward(s, g, mean, rstd)
        else:
            ctx.save_for_backward(res, y, yb, g, mean, rstd)
        ctx.keep_sum = keep_sum
        ctx.has_ybias = ybias is not None
        ctx.pdt = (weight.dtype, bias.dtype, ybias.dtype if ybias is not None else None)
        if keep_sum:
            return s, h
        dummy = h.new_empty(0)
        ctx.mark_non_differentiable(dummy)
        return dummy, h

    @staticmethod
    def backward(ctx, ds, dh):
        if ctx.keep_sum:
            s, g, mean, rstd = ctx.saved_tensors
            dadd = ds.contiguous() if ds is not None else None
            dx, dg, db, dsum = layernorm_bwd_raw(dh.contiguous(), s, None, None, g, mean, rstd, dadd, ctx.has_ybias)
        else:
            res, y, yb, g, mean, rstd = ctx.saved_tensors
            dx, dg, db, dsum = layernorm_bwd_raw(dh.contiguous(), res, y, yb, g, mean, rstd, None, ctx.has_ybias)
        dyb = dsum.to(ctx.pdt[2]) if ctx.has_ybias else None
        return dx, dx, dyb, dg.to(ctx.pdt[0]), db.to(ctx.pdt[1]), None, None


# Residual adds in GEMM epilogues (LVL_EPI_BIAS_RESIDUAL; LAVILA_RESIDUAL_EPILOGUE=0 restores the composed form): the space
# attention's output projection and the MLP's fc2 add the residual stream in their epilogue -- in f32, before the one
# rounding of the sum -- and the LayerNorm behind reads the sum, instead of the GEMM writing y and a fused add + LayerNorm
# reading res and y: one [rows, D] pass less per site. Same-box A/B: 171.56 -> 170.96 ms per step.
RESIDUAL_EPILOGUE = os.environ.get('LAVILA_RESIDUAL_EPILOGUE', '1') != '0'


class _LinearResidualLayerNormFn(torch.autograd.Function):
    """(s, h) = (res + x W^T + b, LayerNorm(s)): `x1 = x + attn(norm1(t)); h2 = norm2(x1)` of SpaceTimeBlock
    (timesformer.py:192-196) as one GEMM with the residual epilogue + one plain LayerNorm pass. Backward: the LayerNorm
    backward kernel adds the stream's own gradient (dadd) and leaves ds, whose column sums are the bias gradient; ds is
    the residual's gradient AND the GEMM's output gradient (input gradient: lvl_linear_tn on the transposed weight copy,
    weight gradient: lvl_linear_wgrad)."""

    @staticmethod
    def forward(ctx, x, weight, lbias, res, gamma, beta, eps, xtoken=None):
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        res2 = res.reshape(-1, res.shape[-1])
        res2 = res2 if res2.is_contiguous() else res2.contiguous()
        w, wt = weight_copies(weight)
        s = linear_tn_raw(x2, w, _f32(lbias), C.EPI_BIAS_RESIDUAL, aux_in=res2)
        g = _f32(gamma)
        h, _, mean, rstd = layernorm_fwd_raw(s, None, None, g, _f32(beta), eps, False)
        ctx.has_token = xtoken is not None
        ctx.save_for_backward(x2, wt, s, g, mean, rstd, *((weight,) if ctx.has_token else ()))
        ctx.meta = (weight.dtype, None if lbias is None else lbias.dtype, gamma.dtype, beta.dtype, x.shape)
        return s.reshape(res.shape), h.reshape(res.shape)

    @staticmethod
    def backward(ctx, ds, dh):
        saved = ctx.saved_tensors             # ONE access: torch.utils.checkpoint's unpack hook refuses a second one
        x2, wt, s, g, mean, rstd = saved[:6]
        wdt, bdt, gdt, betadt, xshape = ctx.meta
        dh2 = dh.reshape(s.shape)
        dadd = None if ds is None else ds.reshape(s.shape).contiguous()
        want_tok = ctx.has_token and ctx.needs_input_grad[7]
        dsum, dg, dbeta, dcol = layernorm_bwd_raw(dh2.contiguous(), s, None, None, g, mean, rstd, dadd,
                                                  bdt is not None or want_tok)
        with torch.autocast('cuda', enabled=False):
            dx = linear_tn_raw(dsum, wt, None, C.EPI_BIAS).reshape(xshape) if ctx.needs_input_grad[0] else None
            dw = _wgrad(dsum, x2, wdt) if ctx.needs_input_grad[1] else None
            # token rule: column sums of dx = (column sums of dsum) . W  (dx = dsum W)
            dtok = vec_mat(dcol, saved[6]) if want_tok else None
        db = dcol.to(bdt) if (bdt is not None and ctx.needs_input_grad[2]) else None
        return (dx, dw, db, dsum.reshape(ds.shape if ds is not None else dh.shape), dg.to(gdt), dbeta.to(betadt), None,
                dtok)


def linear_residual_layer_norm(x, weight, lbias, res, gamma, beta, eps, xtoken=None):
    """(s, h) = (res + Linear(x), LayerNorm(s)) through the GEMM's residual epilogue, or None when the shapes / dtypes are
    not the benched bf16 configuration (the caller then composes linear + add_layer_norm as before). xtoken: the
    column-sum token of x (see COLSUM_TOKENS)."""
    x, res = _act(x), lowp(res)
    rows = x.numel() // x.shape[-1]
    n_out, n_in = weight.shape
    if not (RESIDUAL_EPILOGUE and x.dtype == torch.bfloat16 and res.dtype == torch.bfloat16 and x.is_cuda
            and _tn_ok(rows, n_out, n_in) and _tn_ok(rows, n_in, n_out)):
        return None
    if xtoken is not None and weight.dtype != torch.float32:
        xtoken = None
    s, h = _LinearResidualLayerNormFn.apply(x, weight, lbias, res, gamma, beta, eps, xtoken)
    return s, _narrow(h)


class _MlpResidualLayerNormFn(torch.autograd.Function):
    """(s, h) = (res + fc2(QuickGELU(fc1(x) + b1)) + b2, LayerNorm(s)): `x = x + mlp(norm2(x))` of one block and the first
    LayerNorm of the NEXT consumer (timesformer.py:196 / 183; openai_model.py:200 / 199) -- _MlpFn with the residual
    epilogue on its second GEMM, then one plain LayerNorm pass. Backward = LayerNorm backward (adds the stream's own
    gradient, leaves ds and its column sums = d b2), then _MlpFn's backward on ds."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res, gamma, beta, eps):
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        res2 = res.reshape(-1, res.shape[-1])
        res2 = res2 if res2.is_contiguous() else res2.contiguous()
        w1b, w1t = weight_copies(w1)
        w2b, w2t = weight_copies(w2)
        ctx.deriv = GELU_DERIV
        a, u = linear_tn_raw(x2, w1b, _f32(b1), C.EPI_BIAS_QUICKGELU_DERIV if ctx.deriv else C.EPI_BIAS_QUICKGELU)
        s = linear_tn_raw(a, w2b, _f32(b2), C.EPI_BIAS_RESIDUAL, aux_in=res2)
        g = _f32(gamma)
        h, _, mean, rstd = layernorm_fwd_raw(s, None, None, g, _f32(beta), eps, False)
        ctx.save_for_backward(x2, u, a, w1t, w2t, s, g, mean, rstd)
        ctx.meta = (w1.dtype, None if b1 is None else b1.dtype, w2.dtype, None if b2 is None else b2.dtype, gamma.dtype,
                    beta.dtype, x.shape)
        return s.reshape(res.shape), h.reshape(res.shape)

    @staticmethod
    def backward(ctx, ds, dh):
        x2, u, a, w1t, w2t, s, g, mean, rstd = ctx.saved_tensors
        w1dt, b1dt, w2dt, b2dt, gdt, betadt, xshape = ctx.meta
        dadd = None if ds is None else ds.reshape(s.shape).contiguous()
        dsum, dg, dbeta, dcol = layernorm_bwd_raw(dh.reshape(s.shape).contiguous(), s, None, None, g, mean, rstd, dadd,
                                                  b2dt is not None)
        with torch.autocast('cuda', enabled=False):
            du, db1 = linear_tn_raw(dsum, w2t, None, C.EPI_MUL_AUX_COLSUM if ctx.deriv else C.EPI_QUICKGELU_BWD,
                                    aux_in=u)
            dw2 = _wgrad(dsum, a, w2dt) if ctx.needs_input_grad[3] else None
            dx = linear_tn_raw(du, w1t, None, C.EPI_BIAS).reshape(xshape) if ctx.needs_input_grad[0] else None
            dw1 = _wgrad(du, x2, w1dt) if ctx.needs_input_grad[1] else None
        return (dx, dw1, db1.to(b1dt) if (b1dt is not None and ctx.needs_input_grad[2]) else None, dw2,
                dcol.to(b2dt) if (b2dt is not None and ctx.needs_input_grad[4]) else None,
                dsum.reshape(dh.shape), dg.to(gdt), dbeta.to(betadt), None)


def mlp_residual_layer_norm(x, w1, b1, w2, b2, res, gamma, beta, eps):
    """(s, h) = (res + Mlp(x), LayerNorm(s)) with the residual add in fc2's GEMM epilogue, or None when the configuration
    is not the benched bf16 one (the caller then composes mlp_quickgelu + add_layer_norm)."""
    x, res = _act(x), lowp(res)
    rows = x.numel() // x.shape[-1]
    if not (RESIDUAL_EPILOGUE and x.dtype == torch.bfloat16 and res.dtype == torch.bfloat16 and x.is_cuda
            and b1 is not None and _tn_ok(rows, w1.shape[0], w1.shape[1]) and _tn_ok(rows, w1.shape[1], w1.shape[0])
            and _tn_ok(rows, w2.shape[0], w2.shape[1]) and _tn_ok(rows, w2.shape[1], w2.shape[0])):
        return None
    s, h = _MlpResidualLayerNormFn.apply(x, w1, b1, w2, b2, res, gamma, beta, eps)
    return s, _narrow(h)


class _AddLayerNormPassFn(torch.autograd.Function):
    """(res, h) = (res, LN(res + y + ybias)): the sum is not materialised and `res` is handed through, so that a
    SECOND consumer of res (SpaceTimeBlock: x feeds both x + time_out and x + space_out, timesformer.py:183-196) can
    take it from here. Its gradient then arrives in THIS node and is added inside the LayerNorm backward kernel
    (dx = dx_plain + d_res_out) instead of by a separate full-size add of the autograd engine."""

    @staticmethod
    def forward(ctx, res, y, ybias, weight, bias, eps, ytoken=None):
        res, y = res.contiguous(), y.contiguous()
        yb, g, b = _f32(ybias), _f32(weight), _f32(bias)
        h, _, mean, rstd = layernorm_fwd_raw(res, y, yb, g, b, eps, False)
        ctx.save_for_backward(res, y, yb, g, mean, rstd)
        ctx.has_ybias = ybias is not None
        ctx.has_token = ytoken is not None
        ctx.pdt = (weight.dtype, bias.dtype, ybias.dtype if ybias is not None else None)
        return res.view_as(res), h

    @staticmethod
    def backward(ctx, dres, dh):
        res, y, yb, g, mean, rstd = ctx.saved_tensors
        want_sum = ctx.has_ybias or (ctx.has_token and ctx.needs_input_grad[6])     # column sums of dy
        if dres is None:
            dx, dg, db, dsum = layernorm_bwd_raw(dh.contiguous(), res, y, yb, g, mean, rstd, None, want_sum)
            dy = dx
        else:
            dx, dg, db, dsum, dy = layernorm_bwd_raw(dh.contiguous(), res, y, yb, g, mean, rstd, dres.contiguous(),
                                                     want_sum, want_plain=True)
        dyb = dsum.to(ctx.pdt[2]) if ctx.has_ybias else None
        dtok = dsum if (ctx.has_token and ctx.needs_input_grad[6]) else None      # token rule: sum_rows(dy) itself
        return dx, dy, dyb, dg.to(ctx.pdt[0]), db.to(ctx.pdt[1]), None, dtok


def add_layer_norm_pass(res, y, ybias, weight, bias, eps, ytoken=None):
    """Returns (res_again, h) with h = LayerNorm(res + y (+ ybias)); use res_again for the next consumer of res.
    ytoken: the column-sum token of y (see COLSUM_TOKENS)."""
    res = lowp(res)
    if y.dtype != res.dtype:
        y, ytoken = y.to(res.dtype), None
    r, h = _AddLayerNormPassFn.apply(res, y, ybias, weight, bias, eps, ytoken)
    return r, _narrow(h)


def add_layer_norm(res, y, ybias, weight, bias, eps, keep_sum=True):
    """Returns (s, h) with s = res + y (+ ybias) and h = LayerNorm(s). With keep_sum=False s is None."""
    res = lowp(res)
    if y.dtype != res.dtype:
        y = y.to(res.dtype)
    s, h = _AddLayerNormFn.apply(res, y, ybias, weight, bias, eps, keep_sum)
    return (s if keep_sum else None), _narrow(h)


# --------------------------------------------------------------------------------------------------
# bias + QuickGELU
# --------------------------------------------------------------------------------------------------
class _BiasQuickGELUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, bias):
        u = u.contiguous()
        b = _f32(bias)
        C.require_device(u, b)
        rows, cols = _rows_cols(u)
        a = torch.empty_like(u)
        C.check(C.lib().lvl_bias_quickgelu_fwd(C.ptr(u), C.ptr(b), C.ptr(a), rows, cols, C.dtype_code(u),
                                               C.stream_ptr()), 'lvl_bias_quickgelu_fwd')
        ctx.save_for_backward(u, b)
        ctx.bdt = bias.dtype if bias is not None else None
        return a

    @staticmethod
    def backward(ctx, da):
        u, b = ctx.saved_tensors
        da = da.contiguous()
        rows, cols = _rows_cols(u)
        du = torch.empty_like(u)
        dbias = torch.empty(cols, dtype=torch.float32, device=u.device) if b is not None else None
        ws = C.workspace('bias_quickgelu_bwd', rows, cols, u.device) if b is not None else None
        C.check(C.lib().lvl_bias_quickgelu_bwd(C.ptr(da), C.ptr(u), C.ptr(b), C.ptr(du), C.ptr(dbias), C.ptr(ws),
                                               rows, cols, C.dtype_code(u), C.stream_ptr()),
                'lvl_bias_quickgelu_bwd')
        return du, (dbias.to(ctx.bdt) if b is not None else None)


def bias_quick_gelu(u, bias=None):
    """(u + bias) * sigmoid(1.702 (u + bias))"""
    return _BiasQuickGELUFn.apply(lowp(u), bias)


# --------------------------------------------------------------------------------------------------
# patch gather + token assembly
# --------------------------------------------------------------------------------------------------
def patchify(video: torch.Tensor, patch: int, dtype: torch.dtype, frame_major: bool = False) -> torch.Tensor:
    """[B,C,F,H,W] f32 (or, frame_major, [B,F,C,H,W]) -> [B, F*N, C*P*P]; both layouts are read in place (no gradient
    w.r.t. pixels: the reference never needs one)."""
    video = video.detach()
    if video.dtype != torch.float32:
        video = video.float()          # fp16 clips of the --use-half eval recipes
    video = video.contiguous()
    C.require_device(video)
    if frame_major:
        B, Fr, Ch, H, W = video.shape
    else:
        B, Ch, Fr, H, W = video.shape
    N = (H // patch) * (W // patch)
    out = torch.empty(B, Fr * N, Ch * patch * patch, dtype=dtype, device=video.device)
    C.check(C.lib().lvl_patchify(C.ptr(video), C.ptr(out), B, Ch, Fr, H, W, patch, int(frame_major),
                                 C.dtype_code(out), C.stream_ptr()), 'lvl_patchify')
    return out


EMBED_BWD_KERNEL = os.environ.get('LAVILA_EMBED_BWD_KERNEL', '1') != '0'      # 0: framework reductions (A/B)


class _EmbedTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pe, cls_token, pos_embed, temporal_embed, frames, n_per_frame):
        pe = pe.contiguous()
        B, FN, D = pe.shape
        cls, pos, tem = _f32(cls_token).reshape(-1), _f32(pos_embed).reshape(-1, D), _f32(temporal_embed).reshape(-1, D)
        C.require_device(pe, cls, pos, tem)
        x = torch.empty(B, 1 + FN, D, dtype=pe.dtype, device=pe.device)
        C.check(C.lib().lvl_embed_tokens_fwd(C.ptr(pe), C.ptr(cls), C.ptr(pos), C.ptr(tem), C.ptr(x), B, frames,
                                             n_per_frame, D, C.dtype_code(pe), C.stream_ptr()),
                'lvl_embed_tokens_fwd')
        ctx.dims = (B, frames, n_per_frame, D, temporal_embed.shape[1])
        ctx.pdt = (cls_token.dtype, pos_embed.dtype, temporal_embed.dtype)
        return x

    @staticmethod
    def backward(ctx, dx):
        B, Fr, N, D, num_frames = ctx.dims
        if EMBED_BWD_KERNEL and dx.is_cuda and dx.dtype in (torch.bfloat16, torch.float32) and D % 8 == 0 and B > 0:
            # one pass over dx (lvl_embed_tokens_bwd) instead of a float32 copy of it and three framework reductions
            dx = dx.contiguous()
            dpos = torch.empty(1, N + 1, D, dtype=torch.float32, device=dx.device)
            dtem = torch.empty(1, num_frames, D, dtype=torch.float32, device=dx.device)
            ws = torch.empty(int(C.lib().lvl_embed_tokens_bwd_ws(Fr, N, D)), dtype=torch.float32, device=dx.device)
            C.check(C.lib().lvl_embed_tokens_bwd(C.ptr(dx), C.ptr(dpos), C.ptr(dtem), C.ptr(ws), B, Fr, N, D, num_frames,
                                                 C.dtype_code(dx), C.stream_ptr()), 'lvl_embed_tokens_bwd')
            # d cls_token is its own tensor, NOT a view of dpos: GradScaler.unscale_ / clip_grad_norm_ scale gradients in
            # place, and two parameters sharing gradient storage would be scaled twice (caught by test_grad_scaler_step_*)
            return (dx[:, 1:], dpos[:, 0].reshape(1, 1, D).to(ctx.pdt[0], copy=True), dpos.to(ctx.pdt[1]),
                    dtem.to(ctx.pdt[2]), None, None)
        body = dx[:, 1:].reshape(B, Fr, N, D).float()
        d0 = dx[:, 0].float().sum(0)                                    # cls row: d cls_token = d pos[0]
        dpos = torch.cat([d0[None], body.sum((0, 1))], 0)[None]         # [1, N+1, D]
        dtem = dx.new_zeros((1, num_frames, D), dtype=torch.float32)
        dtem[0, :Fr] = body.sum((0, 2))
        return (dx[:, 1:], d0.reshape(1, 1, D).to(ctx.pdt[0]), dpos.to(ctx.pdt[1]), dtem.to(ctx.pdt[2]),
                None, None)


def embed_tokens(pe, cls_token, pos_embed, temporal_embed, frames, n_per_frame):
    return _EmbedTokensFn.apply(lowp(pe), cls_token, pos_embed, temporal_embed, frames, n_per_frame)


TEXT_EMBED_KERNEL = os.environ.get('LAVILA_TEXT_EMBED_KERNEL', '1') != '0'


class _TextEmbedFn(torch.autograd.Function):
    """x = token_embedding(text) + positional_embedding[:L] (models.py:152-153) as one gather kernel, and its backward as
    a sort-free deterministic segmented sum (csrc/text_embed.hip): torch's embedding_dense_backward takes a rocPRIM radix
    sort above 3072 token rows, whose histogram memsets turn into memset NODES of a replayed hipGraph."""

    @staticmethod
    def forward(ctx, text, table, pos, out_dtype):
        B, L = text.shape
        V, W = table.shape
        tab, p = _f32(table), _f32(pos)
        C.require_device(tab, p)
        x = torch.empty(B, L, W, dtype=out_dtype, device=table.device)
        C.check(C.lib().lvl_text_embed_fwd(C.ptr(text), text.stride(0), C.ptr(tab), C.ptr(p), C.ptr(x), B, L, W, V,
                                           C.dtype_code(x), C.stream_ptr()), 'lvl_text_embed_fwd')
        ctx.text = text                       # an index tensor (no gradient, not an output): kept as a plain attribute
        ctx.meta = (V, W, pos.shape[0], table.dtype, pos.dtype)
        return x

    @staticmethod
    def backward(ctx, dx):
        text = ctx.text
        B, L = text.shape
        V, W, ctx_len, tdt, pdt = ctx.meta
        dx = dx.contiguous()
        dtable = torch.empty(V, W, dtype=torch.float32, device=dx.device)
        dpos = torch.empty(ctx_len, W, dtype=torch.float32, device=dx.device)
        ws = torch.empty(int(C.lib().lvl_text_embed_bwd_ws(B, L, V)), dtype=torch.int32, device=dx.device)
        C.check(C.lib().lvl_text_embed_bwd(C.ptr(dx), C.ptr(text), text.stride(0), C.ptr(dtable), C.ptr(dpos), C.ptr(ws),
                                           B, L, W, V, ctx_len, C.dtype_code(dx), C.stream_ptr()), 'lvl_text_embed_bwd')
        return None, dtable.to(tdt), dpos.to(pdt), None


def text_embed(text, table, pos, out_dtype):
    """CLIP.encode_text's first two lines on the own kernels when they apply (a device int64 [B, L] view whose rows are
    contiguous, float32 table / positions of a width that is a multiple of 8 and at most 2048); None otherwise (the caller
    keeps nn.Embedding)."""
    if not (TEXT_EMBED_KERNEL and text.is_cuda and text.dtype == torch.int64 and text.dim() == 2 and text.stride(1) == 1
            and text.shape[0] > 0 and text.shape[1] > 0
            and table.dtype == torch.float32 and pos.dtype == torch.float32 and table.is_contiguous()
            and pos.is_contiguous() and table.shape[1] % 8 == 0 and table.shape[1] <= 2048
            and pos.shape[0] >= text.shape[1] and out_dtype in (torch.float32, torch.bfloat16)):
        return None
    return _TextEmbedFn.apply(text, table, pos, out_dtype)


# --------------------------------------------------------------------------------------------------
# attention cores
# --------------------------------------------------------------------------------------------------
def set_fp8_qk(on: bool):
    """BASELINE configs[3]'s "fp8 MFMA QK^T path": the streaming space kernels (groups of more than 288 keys, bf16) compute
    their scores with the fp8 matrix instruction on e4m3-rounded q / k fragments, forward and backward consistently
    (include/lavila_hip.h: lvl_set_fp8_qk). Off by default; LAVILA_FP8_QK=1 in the environment is the same switch."""
    C.check(C.lib().lvl_set_fp8_qk(1 if on else 0), 'lvl_set_fp8_qk')


def _qkv_bias_grad(dqkv, dout, dtype):
    """d(bias) of the qkv Linear that feeds an attention core = column sums of dqkv, without reading the k and v
    thirds: every softmax row sums to 1, so sum_rows(dv) = sum_rows(dout) exactly, and the scores do not change when a
    constant vector is added to every key, so sum_rows(dk) = 0. Only the q third is reduced from dqkv itself
    (2/3 of the bytes of the plain reduction saved on q|k|v, 1/3 read back from dout)."""
    D = dout.shape[-1]
    rows = dout.numel() // D
    db = torch.empty(3 * D, dtype=torch.float32, device=dqkv.device)
    ws = C.workspace('qkv_bias_grad', rows, D, dqkv.device)
    C.check(C.lib().lvl_qkv_bias_grad(C.ptr(dqkv), C.ptr(dout), C.ptr(db), C.ptr(ws), rows, D, C.dtype_code(dqkv),
                                      C.stream_ptr()), 'lvl_qkv_bias_grad')
    return db.to(dtype)


class _DividedAttnFn(torch.autograd.Function):
    """`bias` (optional): the bias of the qkv Linear that produced `qkv` (already added there; the caller hands the
    Linear a detached copy). It takes no part in the forward; its gradient comes out of the backward call itself
    (lvl_divided_attn_bwd_bias: q third from the backward kernels' own column sums where they have the rider, k third 0,
    v third from the output's column-sum token when the consumer supplies it -- see COLSUM_TOKENS -- else from dout).
    want_token: also return the column-sum token of `out` (for a consumer that is an own token-aware Function)."""

    @staticmethod
    def forward(ctx, qkv, bias, frames, n_per_frame, heads, mode, want_token=False):
        qkv = qkv.contiguous()
        C.require_device(qkv)
        B, T, D3 = qkv.shape
        D = D3 // 3
        if D != heads * 64 or T != 1 + frames * n_per_frame:
            raise C.HipExtensionError(f'divided attention: qkv {tuple(qkv.shape)} inconsistent with heads={heads} '
                                      f'(head dim must be 64), frames={frames}, patches/frame={n_per_frame}')
        if qkv.dtype == torch.bfloat16 and not C.lib().lvl_attention_fast_path(mode, frames, n_per_frame, heads):
            warn_once(('attn', mode, frames, n_per_frame), f'divided attention ({"time" if mode else "space"}, {frames} '
                      f'frames x {n_per_frame} patches) has no MFMA kernel yet and runs on the generic (slow) kernels')
        out, lse = divided_attn_fwd_raw(qkv, frames, n_per_frame, heads, mode)
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (B, frames, n_per_frame, heads, mode, None if bias is None else bias.dtype)
        ctx.want_token = bool(want_token)
        # an unused token must arrive in backward as None (= "nobody computed the column sums"), not as zeros
        ctx.set_materialize_grads(False)
        if want_token:
            return out, out.new_zeros(D, dtype=torch.float32)      # the token's VALUE is never read; zeros, not uninitialised memory (hooks, anomaly mode)
        return out

    @staticmethod
    def backward(ctx, dout, dtoken=None):
        qkv, out, lse = ctx.saved_tensors
        B, Fr, N, H, mode, bdt = ctx.cfg
        if dout is None:                        # the output itself was not used: zero gradient, no token either
            dout, dtoken = torch.zeros_like(out), None
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        ws = C.workspace('divided_attn_bwd', B * H, 1 + Fr * N, qkv.device)
        if bdt is not None and ctx.needs_input_grad[1]:
            D = H * 64
            db = torch.empty(3 * D, dtype=torch.float32, device=qkv.device)
            n2 = int(C.lib().lvl_divided_attn_bwd_bias_ws(B, Fr, N, H, mode, C.dtype_code(qkv)))
            ws2 = torch.empty(max(n2, 4), dtype=torch.float32, device=qkv.device)
            tok = dtoken.float().contiguous() if dtoken is not None else None
            C.require_device(tok)
            C.check(C.lib().lvl_divided_attn_bwd_bias(C.ptr(qkv), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv),
                                                      C.ptr(ws), C.ptr(tok), C.ptr(db), C.ptr(ws2), B, Fr, N, H, mode,
                                                      C.dtype_code(qkv), C.stream_ptr()), 'lvl_divided_attn_bwd_bias')
            return dqkv, db.to(bdt), None, None, None, None, None
        C.check(C.lib().lvl_divided_attn_bwd(C.ptr(qkv), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv), C.ptr(ws),
                                             B, Fr, N, H, mode, C.dtype_code(qkv), C.stream_ptr()),
                'lvl_divided_attn_bwd')
        return dqkv, None, None, None, None, None, None


def divided_attn_fwd_raw(qkv, frames, n_per_frame, heads, mode):
    """One lvl_divided_attn_fwd call: (out [B,T,D], lse [B,H,T])."""
    B, T, D3 = qkv.shape
    out = torch.empty(B, T, D3 // 3, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, heads, T, dtype=torch.float32, device=qkv.device)
    ws = C.workspace('divided_attn_fwd', B * heads, T, qkv.device)
    C.check(C.lib().lvl_divided_attn_fwd(C.ptr(qkv), C.ptr(out), C.ptr(lse), C.ptr(ws), B, frames, n_per_frame,
                                         heads, mode, C.dtype_code(qkv), C.stream_ptr()), 'lvl_divided_attn_fwd')
    return out, lse


def divided_attention(qkv, frames, n_per_frame, heads, mode, bias=None, want_token=False):
    """mode: 'space' | 'time'. qkv [B,T,3D] -> [B,T,D] (timesformer.py:110-140 between the two Linears).
    bias, want_token: see _DividedAttnFn; with want_token -> (out, token), token None when tokens are switched off."""
    m = {'space': C.ATTN_SPACE, 'time': C.ATTN_TIME}[mode]
    if want_token:
        if COLSUM_TOKENS and bias is not None and torch.is_grad_enabled():
            return _DividedAttnFn.apply(lowp(qkv), bias, frames, n_per_frame, heads, m, True)
        return _DividedAttnFn.apply(lowp(qkv), bias, frames, n_per_frame, heads, m), None
    return _DividedAttnFn.apply(lowp(qkv), bias, frames, n_per_frame, heads, m)


class _ClsAttnFn(torch.autograd.Function):
    """out[b] = attention of the cls query over all tokens (lvl_cls_attn_fwd / _bwd). `bias`: the qkv Linear's bias
    (its q third was added to `q`, its k | v thirds to `kv` by the caller's Linears on detached slices); its gradient
    follows from the softmax identities of _qkv_bias_grad: sum_j dk_j = 0, sum_j dv_j = dout, plus sum_b dq."""

    @staticmethod
    def forward(ctx, q, kv, bias, heads):
        q, kv = q.contiguous(), kv.contiguous()
        C.require_device(q, kv)
        B, T, D2 = kv.shape
        D = D2 // 2
        if D != heads * 64 or tuple(q.shape) != (B, D):
            raise C.HipExtensionError(f'cls attention: q {tuple(q.shape)} / kv {tuple(kv.shape)} inconsistent with '
                                      f'heads={heads} (head dim must be 64)')
        out = torch.empty(B, D, dtype=kv.dtype, device=kv.device)
        lse = torch.empty(B, heads, dtype=torch.float32, device=kv.device)
        C.check(C.lib().lvl_cls_attn_fwd(C.ptr(q), C.ptr(kv), C.ptr(out), C.ptr(lse), B, T, heads, C.dtype_code(kv),
                                         C.stream_ptr()), 'lvl_cls_attn_fwd')
        ctx.save_for_backward(q, kv, out, lse)
        ctx.cfg = (heads, q.dtype, None if bias is None else bias.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse = ctx.saved_tensors
        heads, qdt, bdt = ctx.cfg
        B, T, D2 = kv.shape
        dout = dout.contiguous()
        dq = torch.empty(B, D2 // 2, dtype=torch.float32, device=kv.device)
        dkv = torch.empty_like(kv)
        C.check(C.lib().lvl_cls_attn_bwd(C.ptr(q), C.ptr(kv), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dq), C.ptr(dkv),
                                         B, T, heads, C.dtype_code(kv), C.stream_ptr()), 'lvl_cls_attn_bwd')
        db = None
        if bdt is not None and ctx.needs_input_grad[2]:
            zero = dq.new_zeros(D2 // 2)
            db = torch.cat([dq.sum(0), zero, dout.sum(0, dtype=torch.float32)]).to(bdt)
        return dq.to(qdt), dkv, db, None


def cls_attention(q, kv, heads, bias=None):
    """q [B, D] (cls rows), kv [B, T, 2D] -> [B, D]; see _ClsAttnFn."""
    q, kv = lowp(q), lowp(kv)
    if q.dtype != kv.dtype:
        q = q.to(kv.dtype)
    return _ClsAttnFn.apply(q, kv, bias, heads)


class _CausalAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias, heads):
        qkv = qkv.contiguous()
        C.require_device(qkv)
        B, L, D3 = qkv.shape
        D = D3 // 3
        if D != heads * 64:
            raise C.HipExtensionError(f'causal attention: width {D} != heads*64 ({heads} heads)')
        if qkv.dtype == torch.bfloat16 and not C.lib().lvl_attention_fast_path(2, 1, L, heads):
            warn_once(('attn', 2, L), f'causal attention over {L} tokens runs on the generic (slow) kernels')
        out = torch.empty(B, L, D, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, heads, L, dtype=torch.float32, device=qkv.device)
        C.check(C.lib().lvl_causal_attn_fwd(C.ptr(qkv), C.ptr(out), C.ptr(lse), B, L, heads, C.dtype_code(qkv),
                                            C.stream_ptr()), 'lvl_causal_attn_fwd')
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (B, L, heads, None if bias is None else bias.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, L, H, bdt = ctx.cfg
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        ws = C.workspace('causal_attn_bwd', B * H, L, qkv.device)
        C.check(C.lib().lvl_causal_attn_bwd(C.ptr(qkv), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv), C.ptr(ws),
                                            B, L, H, C.dtype_code(qkv), C.stream_ptr()), 'lvl_causal_attn_bwd')
        db = _qkv_bias_grad(dqkv, dout, bdt) if (bdt is not None and ctx.needs_input_grad[1]) else None
        return dqkv, db, None


def causal_attention(qkv, heads, bias=None):
    """bias: the in_proj bias that produced qkv (see _DividedAttnFn)."""
    return _CausalAttnFn.apply(lowp(qkv), bias, heads)


# --------------------------------------------------------------------------------------------------
# contrastive head (raw kernels; the autograd wrapper + collectives live in lavila_amd/loss.py)
# --------------------------------------------------------------------------------------------------
def clip_loss_fwd_raw(img_all, txt_all, scale, B: int, row0: int, want_logits=False):
    """scale: 0-dim/1-elem float32 DEVICE tensor (exp(logit_scale)); no host sync."""
    C.require_device(img_all, txt_all, scale)
    G, E = img_all.shape
    dev = img_all.device
    stats = torch.empty(2, B, 4, dtype=torch.float32, device=dev)
    argmax = torch.empty(2, B, dtype=torch.int32, device=dev)
    logits = torch.empty(2, B, G, dtype=torch.float32, device=dev) if want_logits else None
    C.check(C.lib().lvl_clip_loss_fwd(C.ptr(img_all), C.ptr(txt_all), C.ptr(scale), B, G, E, row0, C.ptr(stats),
                                      C.ptr(argmax), C.ptr(logits), C.dtype_code(img_all), C.stream_ptr()),
            'lvl_clip_loss_fwd')
    return stats, argmax, logits


def clip_loss_bwd_raw(img_all, txt_all, lse_all, scale, upstream, coef: float, B: int, row0: int,
                      rows_only: bool = False):
    C.require_device(img_all, txt_all, lse_all, scale, upstream)
    G, E = img_all.shape
    dimg = torch.empty(B, E, dtype=torch.float32, device=img_all.device)
    dtxt = torch.empty(B, E, dtype=torch.float32, device=img_all.device)
    C.check(C.lib().lvl_clip_loss_bwd(C.ptr(img_all), C.ptr(txt_all), C.ptr(lse_all), C.ptr(scale), C.ptr(upstream),
                                      float(coef), B, G, E, row0, int(rows_only), C.ptr(dimg), C.ptr(dtxt),
                                      C.dtype_code(img_all),
                                      C.stream_ptr()), 'lvl_clip_loss_bwd')
    return dimg, dtxt


def ssl_clip_loss_fwd_raw(img_all, txt_all, ind_all, scales3, B: int, row0: int, want_logits=False):
    """ind_all: [G] int32 device tensor; scales3: [3] float32 device tensor {pseudo, sqrt(pseudo*real), real}."""
    C.require_device(img_all, txt_all, ind_all, scales3)
    G, E = img_all.shape
    dev = img_all.device
    stats = torch.empty(2, B, 8, dtype=torch.float32, device=dev)
    argmax = torch.empty(2, B, dtype=torch.int32, device=dev)
    logits = torch.empty(2, B, G, dtype=torch.float32, device=dev) if want_logits else None
    C.check(C.lib().lvl_ssl_clip_loss_fwd(C.ptr(img_all), C.ptr(txt_all), C.ptr(ind_all), C.ptr(scales3), B, G, E, row0,
                                          C.ptr(stats), C.ptr(argmax), C.ptr(logits), C.dtype_code(img_all),
                                          C.stream_ptr()), 'lvl_ssl_clip_loss_fwd')
    return stats, argmax, logits


def ssl_clip_loss_bwd_raw(img_all, txt_all, ind_all, lse_all, scales3, upstream, coef: float, B: int, row0: int):
    C.require_device(img_all, txt_all, ind_all, lse_all, scales3, upstream)
    G, E = img_all.shape
    dimg = torch.empty(B, E, dtype=torch.float32, device=img_all.device)
    dtxt = torch.empty(B, E, dtype=torch.float32, device=img_all.device)
    C.check(C.lib().lvl_ssl_clip_loss_bwd(C.ptr(img_all), C.ptr(txt_all), C.ptr(ind_all), C.ptr(lse_all),
                                          C.ptr(scales3), C.ptr(upstream), float(coef), B, G, E, row0, C.ptr(dimg),
                                          C.ptr(dtxt), C.dtype_code(img_all), C.stream_ptr()), 'lvl_ssl_clip_loss_bwd')
    return dimg, dtxt
