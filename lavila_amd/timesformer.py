"""MI355X-native TimeSformer video tower behind the reference's own interface.

Mirrors `lavila/models/timesformer.py` of facebookresearch/LaViLa (class names, constructor
signatures, forward()/forward_features() signatures, parameter/state_dict names, error behaviour),
but the execution plan is ours: no rearrange copies, no materialised attention scores, the
BCTHW->BTCHW permute folded into the patch gather, every residual add fused into the LayerNorm
that consumes it, and every Linear layer (qkv, proj, fc1 + bias + QuickGELU, fc2, patch embedding), the
LayerNorms and the divided space-time attention core run by hand-written HIP kernels through the C ABI
(include/lavila_hip.h: lvl_linear_tn / lvl_linear_wgrad / lvl_layernorm_* / lvl_divided_attn_*). No call of
this file reaches a library GEMM in bf16 unless the shape does not tile (ops.warn_once says so). There is
no CPU path: forward() on a CPU tensor raises. fp16 models / inputs (model.half(), eval_zeroshot.py
--use-half) compute in bf16 and hand fp16 back.
"""
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from . import ops


# LAVILA_CLS_LAST=0 computes the whole last block as the reference does (A/B and debugging)
CLS_ONLY_LAST_BLOCK = __import__('os').environ.get('LAVILA_CLS_LAST', '1') != '0'


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _compute_dtype(weight: torch.Tensor) -> torch.dtype:
    """Activation dtype of the tower: the autocast dtype when autocast is on, else the dtype of the weights;
    fp16 (the reference's AMP dtype, or a model.half()) means bf16 (ops.autocast_dtype / ops.lowp)."""
    lp = ops.autocast_dtype()
    if lp is not None:
        return lp
    return torch.bfloat16 if weight.dtype == torch.float16 else weight.dtype


def _like_caller(out, *given):
    """fp16 in (parameters or input) -> fp16 out, as nn.Module.half() callers expect (eval_zeroshot.py:212-261)."""
    if out.dtype == torch.bfloat16 and not torch.is_autocast_enabled() and \
            any(g is not None and g.dtype == torch.float16 for g in given):
        return out.to(torch.float16)
    return out


class LayerNorm(nn.Module):
    """nn.LayerNorm(dim, eps) with the HIP kernel underneath; same parameter names (weight, bias)."""

    def __init__(self, normalized_shape, eps=1e-5):
        super().__init__()
        dim = normalized_shape if isinstance(normalized_shape, int) else normalized_shape[-1]
        self.normalized_shape = (dim,)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)

    def extra_repr(self):
        return f'{self.normalized_shape}, eps={self.eps}'


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm.models.layers.DropPath semantics, timesformer.py:31,161)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * (mask / keep)


class QuickGELUAct(nn.Module):
    """Marker for the fused bias+QuickGELU kernel; calling it standalone applies x*sigmoid(1.702x)."""

    def forward(self, x):
        return ops.bias_quick_gelu(x, None)


class Mlp(nn.Module):
    """fc2(act(fc1(x))) -- timesformer.py:42-58. With QuickGELU (all CLIP_OPENAI_* models) the fc1 bias add
    and the activation run as one HIP kernel on the raw GEMM output."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self._fused_act = type(self.act).__name__ in ('QuickGELU', 'QuickGELUAct')

    def hidden(self, x):
        if self._fused_act:
            return self.drop(ops.bias_quick_gelu(ops.linear(x, self.fc1.weight), self.fc1.bias))
        return self.drop(self.act(ops.linear(x, self.fc1.weight, self.fc1.bias)))

    def forward(self, x):
        return _like_caller(self.drop(ops.linear(self.hidden(x), self.fc2.weight, self.fc2.bias)), x)


class VideoPatchEmbed(nn.Module):
    """Video to patch embedding -- timesformer.py:61-84. `proj` keeps the Conv2d parameter layout
    ([D,3,P,P], bias only when ln_pre=False) so checkpoints interoperate; the computation is the HIP patch
    gather + one GEMM against proj.weight viewed as [D, 3*P*P]."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=8, ln_pre=False):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0]) * num_frames
        self.num_frames = num_frames
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=not ln_pre)

    def tokens_from_bcthw(self, video):
        """[B,C,F,H,W] -> [B, F*N, D] without the BTCHW copy."""
        assert video.shape[2] <= self.num_frames
        return self._tokens(video, frame_major=False)

    def tokens_from_btchw(self, video):
        """[B,F,C,H,W] (what forward_features / forward receive, timesformer.py:79-84,345-348) -> [B, F*N, D]: the
        gather reads this layout in place as well (no permute().contiguous() round trip of the clip)."""
        assert video.shape[1] <= self.num_frames
        return self._tokens(video, frame_major=True)

    def _tokens(self, video, frame_major):
        if self.patch_size[0] != self.patch_size[1]:
            raise NotImplementedError('non-square patches')
        w = self.proj.weight
        patches = ops.patchify(video, self.patch_size[0], _compute_dtype(w), frame_major=frame_major)
        w2 = w.reshape(w.shape[0], -1)
        k = w2.shape[1]
        if k % 64 and patches.is_cuda and w2.shape[0] % 256 == 0:
            # 14 x 14 patches: 3*14*14 = 588 contraction elements. The MFMA GEMMs walk K in blocks of 64: zero columns up
            # to 640 on both operands (exact) keep the patch embedding of the TSF-L/14 towers on lvl_linear_tn /
            # lvl_linear_wgrad; the weight gradient comes back through the pad's slice
            pad = 64 - k % 64
            patches = F.pad(patches, (0, pad))
            if torch.is_grad_enabled() and w.requires_grad:
                w2 = F.pad(w2, (0, pad))        # part of the autograd graph: one pad (and one cast) per training step
            else:
                # inference: ONE padded copy per parameter state, so that ops.weight_copies' cache (keyed on the tensor it
                # is handed) hits on every later call instead of re-padding and re-casting per forward
                key = (w._version, w.data_ptr(), ops._generation, w.dtype)
                hit = _padded_patch_weights.get(self.proj)
                if hit is None or hit[0] != key:
                    hit = _padded_patch_weights[self.proj] = (key, F.pad(w2.detach(), (0, pad)))
                w2 = hit[1]
        return ops.linear(patches, w2, self.proj.bias)

    def forward(self, x):
        """Reference signature: x [B,F,C,H,W] -> [B*F, D, H/P, W/P] (timesformer.py:79-84)."""
        B, Fr, C, H, W = x.shape
        assert Fr <= self.num_frames
        tok = _like_caller(self.tokens_from_btchw(x), x, self.proj.weight)
        gh, gw = H // self.patch_size[0], W // self.patch_size[1]
        return tok.reshape(B * Fr, gh, gw, -1).permute(0, 3, 1, 2)


_padded_patch_weights = __import__('weakref').WeakKeyDictionary()     # Conv2d module -> (state key, zero-padded [D, 640] weight)


class VarAttention(nn.Module):
    """Divided attention layer -- timesformer.py:87-144. qkv/proj are plain Linears; everything between them
    (head split, q scaling, CLS-attends-all, per-frame / per-location grouping, softmax, merge) is one C-ABI
    call (lvl_divided_attn_fwd / _bwd)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.,
                 initialize='random'):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim != 64:
            raise NotImplementedError(f'lavila_amd attention kernels are built for head_dim 64, got {head_dim}')
        if qk_scale is not None and abs(qk_scale - head_dim ** -0.5) > 1e-12:
            raise NotImplementedError('qk_scale override')
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if initialize == 'zeros':
            self.qkv.weight.data.fill_(0)
            self.qkv.bias.data.fill_(0)
            self.proj.weight.data.fill_(1)
            self.proj.bias.data.fill_(0)
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError('attention / projection dropout (never used by the reference configs)')
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)

    @staticmethod
    def _mode(einops_to, einops_dims):
        if einops_to.replace(' ', '') == '(bf)nd':
            return 'space', int(einops_dims['f'])
        if einops_to.replace(' ', '') == '(bn)fd':
            return 'time', int(einops_dims['n'])
        raise NotImplementedError(f'einops pattern {einops_to!r}')

    def core(self, x, mode, frames, n_per_frame, want_token=False):
        """qkv Linear + attention core; returns the pre-projection tensor [B,T,D] (want_token: and its column-sum token,
        ops.COLSUM_TOKENS -- for callers that hand BOTH to a token-aware consumer and use the tensor nowhere else)."""
        bias = self.qkv.bias
        # the GEMM adds the bias; its gradient comes out of the attention backward call (ops._DividedAttnFn),
        # hence the detached copy for the Linear
        qkv = ops.linear(x, self.qkv.weight, None if bias is None else bias.detach())
        return ops.divided_attention(qkv, frames, n_per_frame, self.num_heads, mode, bias=bias, want_token=want_token)

    def forward(self, x, einops_from, einops_to, einops_dims):
        mode, k = self._mode(einops_to, einops_dims)
        patches = x.shape[1] - 1
        frames, n = (k, patches // k) if mode == 'space' else (patches // k, k)
        return _like_caller(ops.linear(self.core(x, mode, frames, n), self.proj.weight, self.proj.bias), x)


class PendingMlp:
    """A block's `x1 + mlp(norm2(x1))` that has not been computed yet: with ops.RESIDUAL_EPILOGUE the MLP's second GEMM adds
    the residual in its epilogue and the NEXT consumer's LayerNorm reads the sum, so the MLP is enqueued by that consumer
    (ops.mlp_residual_layer_norm). Travels in the `pend` slot of the fused residual chain; the fc2 bias in `pend_bias`."""

    __slots__ = ('h', 'mlp')

    def __init__(self, h, mlp):
        self.h, self.mlp = h, mlp

    def materialize(self):
        """The MLP branch as a tensor (fc2's bias still pending), for consumers that want the composed form."""
        m = self.mlp
        return ops.mlp_quickgelu(self.h, m.fc1.weight, m.fc1.bias, m.fc2.weight)


def _close_pending(res, pend, pend_bias, norm):
    """(s, h) = (res + pend + pend_bias, norm(s)) for a tensor or a PendingMlp in the `pend` slot."""
    if isinstance(pend, PendingMlp):
        m = pend.mlp
        fused = ops.mlp_residual_layer_norm(pend.h, m.fc1.weight, m.fc1.bias, m.fc2.weight, pend_bias, res, norm.weight,
                                            norm.bias, norm.eps)
        if fused is not None:
            return fused
        pend = pend.materialize()
    return ops.add_layer_norm(res, pend, pend_bias, norm.weight, norm.bias, norm.eps, keep_sum=True)


class SpaceTimeBlock(nn.Module):
    """timesformer.py:147-198, 'frozen-in-time' wiring:
         t = x + [tanh(alpha)] * timeattn(norm3(x));  x1 = x + attn(norm1(t));  out = x1 + mlp(norm2(x1))
    """

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, time_init='zeros',
                 attention_style='frozen-in-time', is_tanh_gating=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = VarAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                 attn_drop=attn_drop, proj_drop=drop)
        self.timeattn = VarAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                     attn_drop=attn_drop, proj_drop=drop, initialize=time_init)
        if is_tanh_gating:
            self.alpha_timeattn = nn.Parameter(torch.zeros([]))
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.norm3 = norm_layer(dim)
        self.attention_style = attention_style

    def _dropping(self):
        """Stochastic depth or MLP dropout live in this training forward: the branches are then materialised (own
        GEMMs, separate bias / drop passes) instead of riding on the fused chain (timesformer.py:52-58,192-196)."""
        return self.training and ((isinstance(self.drop_path, DropPath) and self.drop_path.drop_prob > 0.) or
                                  self.mlp.drop.p > 0.)

    def chain(self, res, pend, pend_bias, frames, n_per_frame, defer_mlp=False):
        """One block on the fused residual chain.

        The block input is x = res + pend + pend_bias (pend/pend_bias may be None); that add is fused into
        norm3. Returns (x1, y, y_bias) with the block output = x1 + y + y_bias left *pending* so that the
        next consumer (next block's norm3 or the final norm) fuses it too."""
        if self.attention_style != 'frozen-in-time':
            raise NotImplementedError
        n3, n1, n2 = self.norm3, self.norm1, self.norm2
        if pend is None:
            x = res
            h3 = ops.layer_norm(x, n3.weight, n3.bias, n3.eps)
        else:
            x, h3 = _close_pending(res, pend, pend_bias, n3)
        ta, sa = self.timeattn, self.attn
        tok_y = None
        if hasattr(self, 'alpha_timeattn'):
            o_t = ta.core(h3, 'time', frames, n_per_frame)
            y_t, b_t = torch.tanh(self.alpha_timeattn).to(o_t.dtype) * ops.linear(o_t, ta.proj.weight, ta.proj.bias), None
        else:
            # column-sum tokens (ops.COLSUM_TOKENS): o_t -> projection -> fused add + norm1; the backward hands
            # sum_rows(d o_t) to the attention backward, which needs it for the v third of d(qkv bias)
            o_t, tok_o = ta.core(h3, 'time', frames, n_per_frame, want_token=True)
            (y_t, tok_y), b_t = ops.linear_with_token(o_t, ta.proj.weight, tok_o), ta.proj.bias
        # t = x + time_out is never stored; x is handed through so that its second use below sends its gradient into
        # norm1's backward kernel instead of a separate add
        x, h1 = ops.add_layer_norm_pass(x, y_t, b_t, n1.weight, n1.bias, n1.eps, ytoken=tok_y)
        fused = None
        if not self._dropping():
            # ops.RESIDUAL_EPILOGUE: x1 leaves the projection GEMM (residual epilogue), norm2 reads it
            o_s, tok_s = sa.core(h1, 'space', frames, n_per_frame, want_token=True)
            fused = ops.linear_residual_layer_norm(o_s, sa.proj.weight, sa.proj.bias, x, n2.weight, n2.bias, n2.eps,
                                                   xtoken=tok_s)
        else:
            o_s = sa.core(h1, 'space', frames, n_per_frame)
        if fused is not None:
            x1, h2 = fused
        else:
            if self._dropping():
                y_s, b_s = self.drop_path(ops.linear(o_s, sa.proj.weight, sa.proj.bias)), None
            else:
                y_s, b_s = ops.linear(o_s, sa.proj.weight), sa.proj.bias
            x1, h2 = ops.add_layer_norm(x, y_s, b_s, n2.weight, n2.bias, n2.eps, keep_sum=True)
        if self._dropping():
            return x1, self.drop_path(self.mlp(h2)), None
        if self.mlp._fused_act:
            if defer_mlp and ops.RESIDUAL_EPILOGUE and h2.dtype == torch.bfloat16 and x1.dtype == torch.bfloat16:
                return x1, PendingMlp(h2, self.mlp), self.mlp.fc2.bias      # enqueued by the next consumer (_close_pending)
            return x1, ops.mlp_quickgelu(h2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight), self.mlp.fc2.bias
        return x1, ops.linear(self.mlp.hidden(h2), self.mlp.fc2.weight), self.mlp.fc2.bias

    def chain_cls(self, res, pend, pend_bias, frames, n_per_frame):
        """The LAST block when only the cls row of its output is read (`norm(x)[:, 0]`, timesformer.py:377): same
        arithmetic as `chain`, restricted to what reaches that row. Time attention, norm1 and the k | v thirds of the space
        qkv run on every token (the cls query of the space attention reads the keys / values of all of them); the q
        third, the attention itself (lvl_cls_attn_*: one query per head), its output projection, norm2 and the whole MLP
        -- 2/3 of a block's flops -- run on the B cls rows only. Exact, like
        the caption trim of the text tower: the skipped rows feed nothing (their gradient is exactly zero in the
        reference, too). Returns (x1, y, y_bias) of the cls rows, [B, D] each."""
        n3, n1, n2 = self.norm3, self.norm1, self.norm2
        if pend is None:
            x = res
            h3 = ops.layer_norm(x, n3.weight, n3.bias, n3.eps)
        else:
            x, h3 = _close_pending(res, pend, pend_bias, n3)
        ta, sa = self.timeattn, self.attn
        tok_y = None
        if hasattr(self, 'alpha_timeattn'):
            o_t = ta.core(h3, 'time', frames, n_per_frame)
            y_t, b_t = torch.tanh(self.alpha_timeattn).to(o_t.dtype) * ops.linear(o_t, ta.proj.weight, ta.proj.bias), None
        else:
            o_t, tok_o = ta.core(h3, 'time', frames, n_per_frame, want_token=True)
            (y_t, tok_y), b_t = ops.linear_with_token(o_t, ta.proj.weight, tok_o), ta.proj.bias
        x, h1 = ops.add_layer_norm_pass(x, y_t, b_t, n1.weight, n1.bias, n1.eps, ytoken=tok_y)
        # space attention, cls query only: k | v of every token (the last two thirds of the qkv Linear), q of the cls rows
        D = h1.shape[-1]
        w, bias = sa.qkv.weight, sa.qkv.bias
        bq, bkv = (None, None) if bias is None else (bias.detach()[:D], bias.detach()[D:])
        kv = ops.linear(h1, w[D:], bkv)                                          # [B, T, 2D]
        q = ops.linear(h1[:, 0].contiguous(), w[:D], bq)                         # [B, D]
        o_cls = ops.cls_attention(q, kv, sa.num_heads, bias=bias)                # [B, D]
        y_s = ops.linear(o_cls, sa.proj.weight)                                  # [B, D]
        x1, h2 = ops.add_layer_norm(x[:, 0].contiguous(), y_s, sa.proj.bias, n2.weight, n2.bias, n2.eps, keep_sum=True)
        if self.mlp._fused_act:
            return x1, ops.mlp_quickgelu(h2, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight), self.mlp.fc2.bias
        return x1, ops.linear(self.mlp.hidden(h2), self.mlp.fc2.weight), self.mlp.fc2.bias

    def forward(self, x, einops_from_space, einops_to_space, einops_from_time, einops_to_time,
                time_n, space_f, use_checkpoint=False):
        """Reference signature (timesformer.py:173-174); materialises the block output."""
        frames, n = int(space_f), int(time_n)
        if use_checkpoint:
            x1, y, b = checkpoint.checkpoint(self.chain, x, None, None, frames, n, use_reentrant=False)
        else:
            x1, y, b = self.chain(x, None, None, frames, n)
        return _like_caller(x1 + (y if b is None else y + b.to(y.dtype)), x)


class SpaceTimeTransformer(nn.Module):
    """Divided space-time ViT -- timesformer.py:201-390 (same constructor, attributes and state_dict)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, qk_scale=None, representation_size=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., hybrid_backbone=None, norm_layer=None,
                 num_frames=8, time_init='rand', attention_style='frozen-in-time', ln_pre=False,
                 act_layer=nn.GELU, is_tanh_gating=False):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_frames = num_frames
        norm_layer = norm_layer or partial(LayerNorm, eps=1e-6)
        print("######USING ATTENTION STYLE: ", attention_style)
        if hybrid_backbone is not None:
            raise NotImplementedError('hybrid backbone not implemented')
        if drop_rate != 0.:
            raise NotImplementedError('drop_rate != 0 (never used by the reference configs)')
        self.patch_embed = VideoPatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                           embed_dim=embed_dim, num_frames=num_frames, ln_pre=ln_pre)
        num_patches = self.patch_embed.num_patches
        self.patches_per_frame = num_patches // num_frames

        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patches_per_frame + 1, embed_dim))
        self.temporal_embed = nn.Parameter(torch.zeros(1, num_frames, embed_dim))
        self.ln_pre = LayerNorm(embed_dim, eps=1e-5) if ln_pre else None
        self.pos_drop = nn.Dropout(p=drop_rate)

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            SpaceTimeBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                           qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i],
                           norm_layer=norm_layer, time_init=time_init, attention_style=attention_style,
                           act_layer=act_layer, is_tanh_gating=is_tanh_gating)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)

        if representation_size:
            self.num_features = representation_size
            self.pre_logits = nn.Sequential(OrderedDict([('fc', nn.Linear(embed_dim, representation_size)),
                                                         ('act', nn.Tanh())]))
        else:
            self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()

        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        if num_frames == 1:
            self.apply(self._init_weights)

        self.einops_from_space = 'b (f n) d'
        self.einops_to_space = '(b f) n d'
        self.einops_from_time = 'b (f n) d'
        self.einops_to_time = '(b n) f d'

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, LayerNorm)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def _freeze(self, temporal):
        freeze_list = []
        for n, p in self.named_parameters():
            is_temporal = 'temporal_embed' in n or 'timeattn' in n or 'norm3' in n
            if is_temporal == temporal:
                p.requires_grad = False
                freeze_list.append(n)
        print("Freeze the pretrained parts in vision model: {}".format(freeze_list))

    def freeze_spatial_weights(self):
        self._freeze(temporal=False)

    def freeze_temporal_weights(self):
        self._freeze(temporal=True)

    # ------------------------------------------------------------------------------------------------
    def _features_from_tokens(self, tok, frames, use_checkpoint, cls_at_last, after_block=None):
        """tok: [B, F*N, D] patch-embedded tokens (frame-major). after_block: optional (index, callable), called once
        after that block has been enqueued (CLIP.forward starts the text tower from there); an argument of this one
        call, never module state, so concurrent forwards of one or several models cannot see each other's hook."""
        n = self.patches_per_frame
        if tok.shape[1] != frames * n:
            raise ValueError(f'got {tok.shape[1]} patch tokens for {frames} frames; this model was built for '
                             f'{n} patches per frame (img_size / patch_size are fixed at construction)')
        if ops.RESIDUAL_F32 and tok.dtype != torch.float32:
            tok = tok.float()                  # the stream starts in f32, as cat([cls_token f32, half]) does in the reference
        x = ops.embed_tokens(tok, self.cls_token, self.pos_embed, self.temporal_embed, frames, n)
        if self.ln_pre is not None:
            x = ops.layer_norm(x, self.ln_pre.weight, self.ln_pre.bias, self.ln_pre.eps, stream=True)
        x = self.pos_drop(x)
        res, pend, pend_b = x, None, None
        hook = after_block
        last = len(self.blocks) - 1
        for i, blk in enumerate(self.blocks):
            # the last block of a cls-pooled forward only has to produce its cls rows (SpaceTimeBlock.chain_cls)
            fn = blk.chain_cls if (cls_at_last and i == last and CLS_ONLY_LAST_BLOCK and not blk._dropping()
                                   and blk.attention_style == 'frozen-in-time') else blk.chain
            if use_checkpoint:
                if isinstance(pend, PendingMlp):
                    pend = pend.materialize()
                res, pend, pend_b = checkpoint.checkpoint(fn, res, pend, pend_b, frames, n, use_reentrant=False)
            elif fn == blk.chain and i != last:
                # the block's MLP may wait for the next block's norm3 (ops.RESIDUAL_EPILOGUE: residual add in fc2's epilogue)
                res, pend, pend_b = fn(res, pend, pend_b, frames, n, defer_mlp=True)
            else:
                res, pend, pend_b = fn(res, pend, pend_b, frames, n)
            if hook is not None and i == hook[0]:
                hook[1]()
        nm = self.norm
        if cls_at_last:
            # only row 0 of every sample feeds the output: final residual add + LayerNorm on [B, D]
            r0 = res if res.dim() == 2 else res[:, 0].contiguous()
            if pend is None:
                out = ops.layer_norm(r0, nm.weight, nm.bias, nm.eps)
            else:
                p0 = pend if pend.dim() == 2 else pend[:, 0].contiguous()
                _, out = ops.add_layer_norm(r0, p0, pend_b, nm.weight, nm.bias, nm.eps, keep_sum=False)
            return self.pre_logits(out)
        if pend is None:
            return ops.layer_norm(res, nm.weight, nm.bias, nm.eps)
        return ops.add_layer_norm(res, pend, pend_b, nm.weight, nm.bias, nm.eps, keep_sum=False)[1]

    def forward_features(self, x, use_checkpoint=False, cls_at_last=True):
        """Reference signature: x is [B, F, C, H, W] (timesformer.py:345-382; the narrator's entry with
        cls_at_last=False, narrator.py:74). The gather reads this layout in place."""
        with ops.model_forward():
            b, curr_frames, channels, _, _ = x.shape
            tok = self.patch_embed.tokens_from_btchw(x)
            out = self._features_from_tokens(tok, curr_frames, use_checkpoint, cls_at_last)
            return _like_caller(out, x, self.cls_token)

    def forward(self, x, use_checkpoint=False, _after_block=None):
        """x: [B, C, T, H, W] (timesformer.py:384-390); the BCTHW->BTCHW copy is folded into the gather.
        `_after_block` is not part of the reference signature (see _features_from_tokens)."""
        with ops.model_forward():
            tok = self.patch_embed.tokens_from_bcthw(x)
            out = self._features_from_tokens(tok, x.shape[2], use_checkpoint, True, _after_block)
            return self.head(_like_caller(out, x, self.cls_token))
