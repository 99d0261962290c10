"""Tower-side seam of the narrator (BASELINE configs[4] / SURVEY.md section 8f rank 4), MI355X-native:
`VCLM_HF.encode_image` = video tower (all-token features) -> attention pooling onto `num_img_queries` learned queries
-> LayerNorm, i.e. everything of `lavila/models/narrator.py:31-90` that runs BEFORE the GPT-2 decoder, with the
reference's module / parameter names (`img_queries`, `img_attn_pool.{norm.gamma, context_norm.gamma, to_q.weight,
to_kv.weight, to_out.weight}`, `img_attn_pool_norm.gamma`; the `beta` buffers included) so that a reference
`VCLM_*` checkpoint's `visual.*`, `img_queries`, `img_attn_pool*` entries load unchanged.

`CrossAttention` / `LayerNorm` mirror `lavila/models/coca.py:25-131` (same constructor); the pooling core is one
C-ABI call (lvl_mq_cross_attn_fwd), the projections go through ops.linear (own MFMA GEMMs where the widths tile),
the LayerNorms through lvl_layernorm_fwd. Inference only (the narrator row of the scope table is inference): the
pooling core has no backward kernel and says so. The gated-cross-attention GPT-2 decoder and `generate()`
(narrator.py:92-389, gpt2_gated.py) are NOT built: `VCLM_HF.forward` works with any decoder module handed to the
constructor (it only calls it), `generate` raises. This module is deliberately not aliased under
`lavila.models.narrator`, which keeps resolving to the reference's full implementation.
"""
import torch
import torch.nn as nn

from . import _cabi as C
from . import ops
from .timesformer import SpaceTimeTransformer, _like_caller


class LayerNorm(nn.Module):
    """coca.py:25-34: LayerNorm without a learned bias (`gamma` parameter, `beta` buffer of zeros), eps 1e-5."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    def forward(self, x):
        return ops.layer_norm(x, self.gamma, self.beta, 1e-5)


class _MqCrossAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, heads):
        C.require_device(q, kv)
        B, Tk, two_dh = kv.shape
        if two_dh != 128 or q.shape[-1] != heads * 64:
            raise C.HipExtensionError(f'mq_cross_attention: q {tuple(q.shape)} / kv {tuple(kv.shape)}: head dim must be 64')
        nq = q.shape[-2]
        qb = 0 if q.dim() == 2 or q.shape[0] == 1 else nq * heads * 64          # shared queries: batch stride 0
        if q.dim() == 3 and q.shape[0] not in (1, B):
            raise C.HipExtensionError('mq_cross_attention: query batch must be 1 or the context batch')
        out = torch.empty(B, nq, heads * 64, dtype=kv.dtype, device=kv.device)
        C.check(C.lib().lvl_mq_cross_attn_fwd(C.ptr(q), qb, C.ptr(kv), C.ptr(out), B, nq, heads, Tk, C.dtype_code(kv),
                                              C.stream_ptr()), 'lvl_mq_cross_attn_fwd')
        return out

    @staticmethod
    def backward(ctx, dout):
        raise C.HipExtensionError('mq_cross_attention has no backward kernel: the narrator seam is inference-only '
                                  '(run it under torch.no_grad())')


def mq_cross_attention(q, kv, heads):
    """q [B or 1, NQ, heads*64] (or [NQ, heads*64]), kv [B, T, 128] = k | v -> [B, NQ, heads*64] (coca.py:104-120)."""
    q, kv = ops.lowp(q).contiguous(), ops.lowp(kv).contiguous()
    if q.dtype != kv.dtype:
        q = q.to(kv.dtype)
    return _MqCrossAttnFn.apply(q, kv, heads)


class CrossAttention(nn.Module):
    """coca.py:55-131. One key/value head of `dim_head` channels serves all `heads` query heads (to_kv has 2*dim_head
    outputs). `parallel_ff` (the multimodal-layer variant, unused by the narrator's pooling) is not built."""

    def __init__(self, dim, *, context_dim=None, dim_head=64, heads=8, parallel_ff=False, ff_mult=4, norm_context=False):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError(f'lavila_amd attention kernels are built for head_dim 64, got {dim_head}')
        if parallel_ff:
            raise NotImplementedError('CrossAttention(parallel_ff=True) is not on the narrator pooling path')
        self.heads = heads
        self.scale = dim_head ** -0.5
        inner_dim = heads * dim_head
        context_dim = dim if context_dim is None else context_dim
        self.norm = LayerNorm(dim)
        self.context_norm = LayerNorm(context_dim) if norm_context else nn.Identity()
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(context_dim, dim_head * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)
        self.ff = None

    def forward(self, x, context):
        """x: [b, n, dim] queries (or [n, dim]: the same for every context), context: [b, j, context_dim]."""
        shared = x.dim() == 2 or x.shape[0] == 1
        q = ops.linear(self.norm(x), self.to_q.weight)              # computed once when the queries are shared
        kv = ops.linear(self.context_norm(context), self.to_kv.weight)
        out = mq_cross_attention(q if not shared else q.reshape(-1, q.shape[-1]), kv, self.heads)
        return _like_caller(ops.linear(out, self.to_out.weight), context)


class VCLM_HF(nn.Module):
    """narrator.py:31-115: same constructor; encode_image is built on the HIP path, forward() only wires the given
    decoder, generate()/beam search belong to the decoder side and are not built."""

    def __init__(self, vision_width: int, vision_model: nn.Module, text_width: int, text_decoder: nn.Module,
                 num_img_queries=256, dim_head=64, heads=8, **kwargs):
        super().__init__()
        self.vision_width = vision_width
        self.visual = vision_model
        self.text_width = text_width
        self.text_decoder = text_decoder
        self.img_queries = nn.Parameter(torch.empty(num_img_queries, text_width))
        self.img_attn_pool = CrossAttention(dim=text_width, context_dim=vision_width, dim_head=dim_head, heads=heads,
                                            norm_context=True)
        self.img_attn_pool_norm = LayerNorm(text_width)
        self.initialize_parameters()

    def initialize_parameters(self):
        nn.init.normal_(self.img_queries, std=self.text_width ** -0.5)

    def encode_image(self, image, use_checkpoint=False):
        """image [B,C,T,H,W] -> [B, num_img_queries, text_width] (narrator.py:63-90). The reference permutes the clip to
        BTCHW with a copy and the features to BDN and back; here the tower reads BCTHW in place and hands [B, T, D] on."""
        if not isinstance(self.visual, SpaceTimeTransformer):
            raise NotImplementedError('VCLM_HF.encode_image: only the SpaceTimeTransformer tower is built (narrator.py:73-76)')
        with ops.model_forward():
            tok = self.visual.patch_embed.tokens_from_bcthw(image)
            x = self.visual._features_from_tokens(tok, image.shape[2], use_checkpoint, False)   # [B, 1 + F*N, D]
            pooled = self.img_attn_pool(self.img_queries, x)          # queries shared by the batch: projected once
            return _like_caller(self.img_attn_pool_norm(pooled), image, self.img_queries)

    def forward(self, image, text, mask=None, use_checkpoint=False, norm_embed=False):
        """narrator.py:92-110 around whatever decoder the constructor was given."""
        if self.text_decoder is None:
            raise NotImplementedError('VCLM_HF.forward needs a text decoder; the gated GPT-2 of the reference '
                                      '(gpt2_gated.py) is not built in lavila_amd')
        if use_checkpoint:
            self.text_decoder.gradient_checkpointing_enable()
        else:
            self.text_decoder.gradient_checkpointing_disable()
        text, labels = text[:, :-1], text[:, 1:]
        image_tokens = self.encode_image(image, use_checkpoint=use_checkpoint)
        logits = self.text_decoder(text.contiguous(), encoder_hidden_states=image_tokens).logits
        return {'text_tokens_logits': logits.permute(0, 2, 1), 'labels': labels}

    def generate(self, *args, **kwargs):
        raise NotImplementedError('narrator decoding (narrator.py:112-389) is outside the built scope: only the '
                                  'tower-side seam (encode_image) runs on the HIP path')
