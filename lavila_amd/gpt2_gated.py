"""The narrator's text decoder -- GPT-2 with gated cross-attention (lavila/models/gpt2_gated.py), MI355X-native,
INFERENCE ONLY (BASELINE configs[4]: captioning).

Module and parameter names are the reference's (`transformer.{wte,wpe,ln_f}`, `transformer.h.{i}.{ln_1, attn.{c_attn,
c_proj}, ln_2, mlp.{c_fc,c_proj}, crossattention.{c_attn,q_attn,c_proj}, ln_cross_attn, mlp_crossattention.{c_fc,c_proj},
ln_2_crossattention, alpha_cattn, alpha_dense}`, `lm_head`; Conv1D weights stay [in, out]; the `attn.bias` /
`attn.masked_bias` mask buffers are kept as state) so that `text_decoder.*` of a reference VCLM checkpoint loads with
strict=True. The modules only HOLD parameters; the computation is laid out for the device instead of module by module:

  * the Conv1D GEMMs run on lvl_linear_tn against a packed inference image of the weights (bf16, [out, in], built once per
    parameter state; the vocabulary is padded to the GEMM's 256-column tiles), f32 models (the parity configuration) on
    the library GEMM against the masters;
  * every residual add (with its tanh(alpha) gate) is fused with the LayerNorm that reads the sum next
    (lvl_gated_add_layernorm) and, while decoding, both are folded into the prologue of the Conv1D that consumes the
    normalised rows (lvl_linear_skinny_ln); the embedding is one gather (lvl_gpt2_embed);
  * teacher-forced forward (`GPT2LMHeadModel.forward`, what VCLM_HF.forward and target scoring call): the causal
    self-attention is lvl_causal_attn_fwd, the cross-attention lvl_cross_attn_rows_fwd over keys / values projected from
    the image tokens ONCE per clip;
  * decoding (`GPT2LMHeadModel.decode_session`, what VCLM_HF.generate drives): ONE new row per caption and step against
    a key/value cache (lvl_decode_self_attn); the position is a device scalar, so a single captured hipGraph is replayed
    for every step. The reference re-runs the whole prefix per token (narrator.py:118-143, `use_cache=False`): same
    numbers, quadratically more work.

What the reference's vendored HF class offers beyond this path (attention / head masks, token types, past_key_values as
arguments, pruning, model parallelism, the other heads) raises NotImplementedError. Training the decoder is not built:
calling it with gradients enabled on parameters that require them raises.
"""
import copy
import os
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _cabi as C
from . import ops

# rows up to which a Conv1D goes to lvl_linear_skinny (one workgroup per 16/32 weight columns) instead of the
# 256x256-tile kernel: a decode step of 64 captions measured 32 us per GEMM on the latter (one CU per 256-column panel);
# the upper end is measured on the teacher-forced pass (rows = captions x positions): the reference's recompute schedule for
# 64 captions x 77 tokens takes 479 / 432 / 345 / 410 ms with the limit at 1024 / 2048 / 8192 / 65536 rows
SKINNY_MAX_ROWS = int(os.environ.get('LAVILA_SKINNY_MAX_ROWS', '8192'))
# rows up to which the residual add + LayerNorm in front of a Conv1D is folded into that GEMM (lvl_linear_skinny_ln)
FUSED_LN_MAX_ROWS = 128

_SIZES = {  # hidden, layers, heads of the published GPT-2 checkpoints (models.py:729,770,914: "gpt2", "gpt2-large", "gpt2-xl")
    'gpt2': (768, 12, 12), 'gpt2-medium': (1024, 24, 16), 'gpt2-large': (1280, 36, 20), 'gpt2-xl': (1600, 48, 25),
}


def gpt2_config(name='gpt2', **overrides):
    """The fields of transformers' GPT2Config that the decoder reads, for the published sizes -- so that the VCLM_*
    constructors work without the hub (there is no network here). A real GPT2Config is accepted everywhere instead."""
    width, layers, heads = _SIZES[name]
    cfg = types.SimpleNamespace(
        vocab_size=50257, n_positions=1024, n_embd=width, n_layer=layers, n_head=heads, n_inner=None,
        activation_function='gelu_new', layer_norm_epsilon=1e-5, initializer_range=0.02, scale_attn_weights=True,
        scale_attn_by_inverse_layer_idx=False, reorder_and_upcast_attn=False, tie_word_embeddings=True,
        add_cross_attention=False, bos_token_id=50256, eos_token_id=50256, use_cache=False)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def augment_gpt2_config(config, cross_attn_freq=1, gated_xattn=True):
    """gpt2_gated.py:84-90."""
    new_config = copy.deepcopy(config)
    new_config.add_cross_attention = True
    new_config.add_cross_attention_freq = cross_attn_freq
    new_config.is_tanh_gating = gated_xattn
    return new_config


def _cfg(config, name, default=None):
    aliases = {'n_embd': 'hidden_size', 'n_layer': 'num_hidden_layers', 'n_head': 'num_attention_heads',
               'n_positions': 'max_position_embeddings'}
    if hasattr(config, name):
        return getattr(config, name)
    if name in aliases and hasattr(config, aliases[name]):
        return getattr(config, aliases[name])
    return default


class Conv1D(nn.Module):
    """transformers.pytorch_utils.Conv1D as gpt2_gated.py:184-188,383-384 uses it: y = x @ weight + bias with
    weight [in, out] (normal std 0.02), bias [out] zeros."""

    def __init__(self, nf, nx):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf))
        self.bias = nn.Parameter(torch.zeros(nf))
        nn.init.normal_(self.weight, std=0.02)


class GPT2Attention(nn.Module):
    """Parameter holder of gpt2_gated.py:149-189 (self-attention: c_attn -> 3*D; cross-attention: q_attn on the text,
    c_attn -> 2*D on the image tokens; c_proj)."""

    def __init__(self, config, is_cross_attention=False, layer_idx=None):
        super().__init__()
        D, P = _cfg(config, 'n_embd'), _cfg(config, 'n_positions')
        self.register_buffer('bias', torch.tril(torch.ones((P, P), dtype=torch.uint8)).view(1, 1, P, P))
        self.register_buffer('masked_bias', torch.tensor(-1e4))
        self.embed_dim = D
        self.num_heads = _cfg(config, 'n_head')
        self.head_dim = D // self.num_heads
        self.is_cross_attention = is_cross_attention
        self.layer_idx = layer_idx
        if is_cross_attention:
            self.c_attn = Conv1D(2 * D, D)
            self.q_attn = Conv1D(D, D)
        else:
            self.c_attn = Conv1D(3 * D, D)
        self.c_proj = Conv1D(D, D)


class GPT2MLP(nn.Module):
    """gpt2_gated.py:379-396: c_fc -> gelu_new (or relu^2 in the cross-attention MLP) -> c_proj."""

    def __init__(self, intermediate_size, config, squared_relu=False):
        super().__init__()
        D = _cfg(config, 'n_embd')
        self.c_fc = Conv1D(intermediate_size, D)
        self.c_proj = Conv1D(D, intermediate_size)
        self.squared_relu = squared_relu


class GPT2Block(nn.Module):
    """gpt2_gated.py:399-419."""

    def __init__(self, config, layer_idx=None):
        super().__init__()
        D = _cfg(config, 'n_embd')
        inner = _cfg(config, 'n_inner') or 4 * D
        eps = _cfg(config, 'layer_norm_epsilon', 1e-5)
        self.ln_1 = nn.LayerNorm(D, eps=eps)
        self.attn = GPT2Attention(config, layer_idx=layer_idx)
        self.ln_2 = nn.LayerNorm(D, eps=eps)
        self.add_cross_attention_freq = _cfg(config, 'add_cross_attention_freq', 1)
        if _cfg(config, 'add_cross_attention', False) and layer_idx % self.add_cross_attention_freq == 0:
            self.crossattention = GPT2Attention(config, is_cross_attention=True, layer_idx=layer_idx)
            self.ln_cross_attn = nn.LayerNorm(D, eps=eps)
            self.mlp_crossattention = GPT2MLP(inner, config, squared_relu=True)
            self.ln_2_crossattention = nn.LayerNorm(D, eps=eps)
            if _cfg(config, 'is_tanh_gating', False):
                self.alpha_cattn = nn.Parameter(torch.zeros([]))
                self.alpha_dense = nn.Parameter(torch.zeros([]))
        self.mlp = GPT2MLP(inner, config)

    @property
    def has_cross(self):
        return hasattr(self, 'crossattention')


class GPT2Model(nn.Module):
    """gpt2_gated.py:726-747 (parameters)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        D = _cfg(config, 'n_embd')
        self.embed_dim = D
        self.wte = nn.Embedding(_cfg(config, 'vocab_size'), D)
        self.wpe = nn.Embedding(_cfg(config, 'n_positions'), D)
        self.h = nn.ModuleList([GPT2Block(config, layer_idx=i) for i in range(_cfg(config, 'n_layer'))])
        self.ln_f = nn.LayerNorm(D, eps=_cfg(config, 'layer_norm_epsilon', 1e-5))
        std = _cfg(config, 'initializer_range', 0.02)
        nn.init.normal_(self.wte.weight, std=std)
        nn.init.normal_(self.wpe.weight, std=std)


class CausalLMOutput:
    """What VCLM_HF reads of transformers' CausalLMOutputWithCrossAttentions: `.logits`, `.loss`, `[0]`."""

    def __init__(self, logits, loss=None):
        self.logits, self.loss, self.past_key_values = logits, loss, None

    def __getitem__(self, i):
        return (((self.loss,) if self.loss is not None else ()) + (self.logits,))[i]


# --------------------------------------------------------------------------------------------------
# device-side image of the parameters + the plan that runs on it
# --------------------------------------------------------------------------------------------------
def _pad_rows(t, multiple):
    n = -t.shape[0] % multiple
    return t if n == 0 else torch.cat([t, t.new_zeros(n, *t.shape[1:])])


class _Pack:
    """Inference image of the decoder's parameters for one compute dtype: bf16 -> every Conv1D as a contiguous [out, in]
    bf16 matrix (what lvl_linear_tn streams), the token table padded to a multiple of 256 rows (it doubles as the
    lm_head operand when tied), position table in bf16, biases / LayerNorm parameters / tanh(alpha) gates in f32.
    f32 (the parity configuration; round 5) -> every Conv1D as the bf16 TERM IMAGES [out, 3 in] of its float32 [out, in]
    matrix (ops.split3 role 1): the products run on the own MFMA kernels in f32-class mode (lvl_linear_skinny_f32c, or
    lvl_linear_tn where it tiles), ~2^-17 relative per product -- no library GEMM on the decoder's float32 path either."""

    def __init__(self, model, dtype):
        tr = model.transformer
        self.dtype = dtype
        self.key = model._param_key()
        self.D = tr.embed_dim
        self.heads = tr.h[0].attn.num_heads
        self.vocab = tr.wte.weight.shape[0]
        self.positions = tr.wpe.weight.shape[0]
        self.eps = float(tr.ln_f.eps)
        lowp = dtype == torch.bfloat16
        f32 = lambda p: p.detach().to(torch.float32).contiguous()
        # float32: every Conv1D width is a multiple of the model width (a multiple of 64): the f32-class kernels take them all
        self.own = (not lowp) and ops.F32_MFMA and self.D % 32 == 0

        def conv(m):      # -> (matrix, bias f32, out, in)
            w = m.weight.detach()
            if lowp:
                return (w.t().to(torch.bfloat16).contiguous(), f32(m.bias), w.shape[1], w.shape[0])
            if self.own:
                return (ops.split3(f32(w.t()), 1), f32(m.bias), w.shape[1], w.shape[0])
            return (f32(w), f32(m.bias), w.shape[1], w.shape[0])

        def ln(m):
            return (f32(m.weight), f32(m.bias))

        def gate(blk, name):
            return torch.tanh(getattr(blk, name).detach().to(torch.float32)).reshape(1) if hasattr(blk, name) else None

        self.blocks = []
        for blk in tr.h:
            e = {'ln_1': ln(blk.ln_1), 'c_attn': conv(blk.attn.c_attn), 'c_proj': conv(blk.attn.c_proj),
                 'ln_2': ln(blk.ln_2), 'fc': conv(blk.mlp.c_fc), 'proj': conv(blk.mlp.c_proj), 'cross': blk.has_cross}
            if blk.has_cross:
                e.update({'ln_x': ln(blk.ln_cross_attn), 'xq': conv(blk.crossattention.q_attn),
                          'xkv': conv(blk.crossattention.c_attn), 'xproj': conv(blk.crossattention.c_proj),
                          'ln_2x': ln(blk.ln_2_crossattention), 'xfc': conv(blk.mlp_crossattention.c_fc),
                          'xfproj': conv(blk.mlp_crossattention.c_proj),
                          'gate_c': gate(blk, 'alpha_cattn'), 'gate_d': gate(blk, 'alpha_dense')})
            self.blocks.append(e)
        self.ln_f = ln(tr.ln_f)
        wte, head = tr.wte.weight.detach(), model.lm_head.weight.detach()
        tied = head.data_ptr() == wte.data_ptr()
        if lowp:
            self.wte = _pad_rows(wte.to(torch.bfloat16), 256)
            self.wpe = tr.wpe.weight.detach().to(torch.bfloat16).contiguous()
            self.head = self.wte if tied else _pad_rows(head.to(torch.bfloat16), 256)
            self.head3 = None
        else:
            self.wte, self.wpe = f32(wte), f32(tr.wpe.weight)
            self.head = self.wte if tied else f32(head)
            self.head3 = ops.split3(_pad_rows(self.head, 256), 1) if self.own else None

    # ---- primitives --------------------------------------------------------------------------------
    def _skinny(self, x2, w, b, act):
        return ops.linear_skinny_raw(x2, w, b, act)

    def gemm(self, x2, entry, act=None):
        """Conv1D (+ activation). bf16: few rows (decoding) -> lvl_linear_skinny with the activation in its epilogue;
        many rows (teacher-forced captions, the image keys / values) -> lvl_linear_tn, activation in place afterwards;
        widths neither kernel tiles -> library GEMM (logged). f32: library GEMM against the [in, out] master."""
        w, b, n_out, n_in = entry
        rows = x2.shape[0]
        if self.dtype == torch.bfloat16:
            if rows <= SKINNY_MAX_ROWS and n_out % 16 == 0 and n_in % 32 == 0:
                return self._skinny(x2, w, b, act)
            if ops._tn_ok(rows, n_out, n_in):
                y = ops.linear_tn_raw(x2, w, b, C.EPI_BIAS)
            else:
                ops.warn_once(('conv1d', n_out, n_in), f'decoder Conv1D [{n_in}->{n_out}] on {rows} rows runs on the '
                              'library GEMM (lvl_linear_tn needs out % 256 == 0 and in % 64 == 0)')
                y = F.linear(x2, w, b.to(torch.bfloat16))
        elif self.own:                                  # float32 decoder, term images: own kernels in f32-class mode
            return ops.linear_f32_rows(x2, w, b, act)
        else:                                           # LAVILA_F32_MFMA=0 (A/B): library GEMM against the [in, out] master
            y = torch.addmm(b, x2, w)
        return y if act is None else self.act(y, act)

    def logits(self, h2):
        """lm_head (no bias, gpt2_gated.py:1010,1139): [rows, D] -> [rows, vocab] (a view of the padded product)."""
        if self.dtype == torch.bfloat16:
            # 50432 x 768 (profiles/r03_skinny_variants.json, in-graph): up to 128 rows lvl_linear_skinny's LDS-staged tiles
            # (17 us; the 256-column-panel kernel 32), beyond that the panel kernel (72 us at 640 rows; tiles 87-120)
            if h2.shape[0] > 128 and ops._tn_ok(h2.shape[0], self.head.shape[0], self.D) and self.head.shape[0] >= 8192:
                return ops.linear_tn_raw(h2, self.head, None, C.EPI_BIAS)[:, :self.vocab]
            if h2.shape[0] <= SKINNY_MAX_ROWS and self.D % 32 == 0:
                return self._skinny(h2, self.head, None, None)[:, :self.vocab]
            if ops._tn_ok(h2.shape[0], self.head.shape[0], self.D):
                return ops.linear_tn_raw(h2, self.head, None, C.EPI_BIAS)[:, :self.vocab]
        elif self.head3 is not None:                    # float32: f32-class mode of the own kernels
            return ops.linear_f32_rows(h2, self.head3)[:, :self.vocab]
        return F.linear(h2, self.head)[:, :self.vocab]

    def embed(self, ids, L, pos_dev=None):
        ids = ids.reshape(-1).contiguous()
        C.require_device(ids)
        out = torch.empty(ids.shape[0], self.D, dtype=self.dtype, device=ids.device)
        C.check(C.lib().lvl_gpt2_embed(C.ptr(ids), C.ptr(self.wte), C.ptr(self.wpe), C.ptr(pos_dev), C.ptr(out),
                                       ids.shape[0], L, self.D, self.vocab, self.positions, C.dtype_code(out),
                                       C.stream_ptr()), 'lvl_gpt2_embed')
        return out

    def add_ln(self, res, y, gate, ln):
        """res <- res + gate * y (in place), returns LayerNorm(res)."""
        h = torch.empty_like(res)
        C.check(C.lib().lvl_gated_add_layernorm(C.ptr(res), C.ptr(y), C.ptr(gate), C.ptr(ln[0]), C.ptr(ln[1]), self.eps,
                                                C.ptr(res) if y is not None else None, C.ptr(h), res.shape[0], self.D,
                                                C.dtype_code(res), C.stream_ptr()), 'lvl_gated_add_layernorm')
        return h

    def act(self, u, which):
        C.check(C.lib().lvl_act_inplace(C.ptr(u), u.numel(), which, C.dtype_code(u), C.stream_ptr()), 'lvl_act_inplace')
        return u

    def cross_attn(self, q, kv, qrep):
        out = torch.empty_like(q)
        C.check(C.lib().lvl_cross_attn_rows_fwd(C.ptr(q), C.ptr(kv), C.ptr(out), q.shape[0], qrep, kv.shape[1],
                                                self.heads, C.dtype_code(q), C.stream_ptr()), 'lvl_cross_attn_rows_fwd')
        return out

    def image_kv(self, enc):
        """crossattention.c_attn on the image tokens (gpt2_gated.py:330), once per clip and cross-attention block:
        [Bc, NQ, D] -> list of [Bc, NQ, 2D] (None for blocks without cross-attention)."""
        enc2 = enc.reshape(-1, enc.shape[-1]).to(self.dtype).contiguous()
        return [self.gemm(enc2, e['xkv']).reshape(enc.shape[0], enc.shape[1], 2 * self.D) if e['cross'] else None
                for e in self.blocks]

    def first_ln(self, i, with_image):
        if i == len(self.blocks):
            return self.ln_f
        e = self.blocks[i]
        return e['ln_x'] if (e['cross'] and with_image) else e['ln_1']

    def gemm_ln(self, x, pend, entry, act=None, fuse=True):
        """Conv1D of LayerNorm(x + gate * y) for the pending (y, gate, ln): -> (product, new residual). Few rows in bf16
        and `fuse`: ONE kernel (lvl_linear_skinny_ln: add, statistics and normalisation in the GEMM's prologue; the new
        residual goes to a fresh buffer because the other column strips still read the old one). Otherwise the fused
        add + LayerNorm kernel (residual updated in place) followed by the GEMM. Measured at 64 captions
        (profiles/r03_narrator_decode_n1.json): launched one by one the folded form is 15 % faster per token step (48
        launches fewer); inside a replayed hipGraph it is 2.5 % SLOWER (the prologue's two barriers sit on the critical
        path of every column strip, where the stand-alone kernel costs 2.5 us once) -- so graph replay keeps them apart."""
        y, gate, ln = pend
        w, b, n_out, n_in = entry
        rows = x.shape[0]
        if (fuse and self.dtype == torch.bfloat16 and rows <= FUSED_LN_MAX_ROWS and n_out % 16 == 0 and n_in % 32 == 0
                and n_in <= 1792):
            out = torch.empty(rows, n_out, dtype=torch.bfloat16, device=x.device)
            new_x = torch.empty_like(x) if y is not None else x
            C.require_device(x, y, w, b)
            C.check(C.lib().lvl_linear_skinny_ln(C.ptr(x), C.ptr(y), C.ptr(gate), C.ptr(ln[0]), C.ptr(ln[1]), self.eps,
                                                 C.ptr(new_x) if y is not None else None, C.ptr(w), C.ptr(b), C.ptr(out),
                                                 rows, n_out, n_in, -1 if act is None else act, C.stream_ptr()),
                    'lvl_linear_skinny_ln')
            return out, new_x
        h = self.add_ln(x, y, gate, ln)
        return self.gemm(h, entry, act), x

    def run(self, x, xkv, qrep, self_attention, fuse_ln=True):
        """The block stack on rows x [rows, D] -> LayerNorm_f of the final residual. xkv: per block image keys / values
        or None (no encoder states: plain GPT-2, gpt2_gated.py:432); `self_attention(i, qkv)` -> [rows, D]. Every
        residual add + LayerNorm is PENDING until the Conv1D that reads it (gemm_ln)."""
        with_image = xkv is not None
        pend = (None, None, self.first_ln(0, with_image))
        for i, e in enumerate(self.blocks):
            if e['cross'] and with_image:
                q, x = self.gemm_ln(x, pend, e['xq'], None, fuse_ln)
                a = self.cross_attn(q, xkv[i], qrep)
                pend = (self.gemm(a, e['xproj']), e['gate_c'], e['ln_2x'])
                u, x = self.gemm_ln(x, pend, e['xfc'], C.ACT_SQRELU, fuse_ln)
                pend = (self.gemm(u, e['xfproj']), e['gate_d'], e['ln_1'])
            qkv, x = self.gemm_ln(x, pend, e['c_attn'], None, fuse_ln)
            a = self_attention(i, qkv)
            pend = (self.gemm(a, e['c_proj']), None, e['ln_2'])
            u, x = self.gemm_ln(x, pend, e['fc'], C.ACT_GELU_NEW, fuse_ln)
            pend = (self.gemm(u, e['proj']), None, self.first_ln(i + 1, with_image))
        return self.add_ln(x, *pend)                     # ln_f feeds lm_head (the 256-column-panel kernel): materialised


class DecodeSession:
    """Key/value-cached decoding of `rows = contexts * seqs_per_context` captions, one token per step.
    step(ids [rows]) -> logits [rows, vocab] of the NEXT token (a view of a buffer that the next step overwrites).
    With graph=True the step is captured once into a hipGraph (position, ids and logits live in fixed device buffers)
    and replayed; the session is bound to the parameter values it was built from (`stale()` tells)."""

    def __init__(self, model, pack, image_tokens, max_length, seqs_per_context=1, graph=True):
        self.model, self.pack = model, pack
        dev = image_tokens.device
        Bc = image_tokens.shape[0]
        self.rows, self.qrep, self.capacity = Bc * seqs_per_context, seqs_per_context, int(max_length)
        if self.capacity > pack.positions:
            raise ValueError(f'decode length {self.capacity} exceeds the decoder\'s {pack.positions} positions')
        self.contexts, self.context_len = Bc, image_tokens.shape[1]
        self.xkv = pack.image_kv(image_tokens)
        self.cache = [torch.zeros(self.rows, self.capacity, 2 * pack.D, dtype=pack.dtype, device=dev) for _ in pack.blocks]
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ids = torch.zeros(self.rows, dtype=torch.int64, device=dev)
        self.steps = 0
        self._logits = None
        self._graph = None
        self.graphed = bool(graph)
        if graph:
            self._capture()

    def stale(self):
        return self.pack.key != self.model._param_key()

    def _self_attention(self, i, qkv):
        out = torch.empty(self.rows, self.pack.D, dtype=qkv.dtype, device=qkv.device)
        C.check(C.lib().lvl_decode_self_attn(C.ptr(qkv), C.ptr(self.cache[i]), C.ptr(self.pos), C.ptr(out), self.rows,
                                             self.capacity, self.pack.heads, C.dtype_code(qkv), C.stream_ptr()),
                'lvl_decode_self_attn')
        return out

    def _run(self):
        p = self.pack
        x = p.embed(self.ids, 1, self.pos)
        h = p.run(x, self.xkv, self.qrep, self._self_attention, fuse_ln=not self.graphed)
        logits = p.logits(h)
        self.pos.add_(1)
        return logits

    def _capture(self):
        """One eager step on a side stream (lazy kernel attributes, allocator warm-up), then the capture. A capture that
        fails (e.g. another thread launching on the device meanwhile) leaves the session on eager launches, loudly."""
        try:
            self._capture_step()
        except RuntimeError as e:                        # hipGraph errors surface as RuntimeError from torch
            import warnings
            warnings.warn(f'DecodeSession: hipGraph capture failed ({e}); decoding with eager launches')
            self._graph, self._logits, self.graphed = None, None, False
            torch.cuda.synchronize()
            self.pos.zero_()

    def _capture_step(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # first launch outside the capture: lazy kernel attributes, allocator warm-up
            self._run()
            self.pos.zero_()
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._logits = self._run()
        self.pos.zero_()

    def reset(self):
        self.pos.zero_()
        self.steps = 0

    def rebind(self, image_tokens):
        """Start over on other clips of the same shape: their keys / values are projected INTO the buffers the captured
        graph reads, the position returns to 0 (cache rows are rewritten before they are read)."""
        if tuple(image_tokens.shape[:2]) != (self.contexts, self.context_len):
            raise ValueError('DecodeSession.rebind: image tokens of another shape need another session')
        for dst, src in zip(self.xkv, self.pack.image_kv(image_tokens)):
            if dst is not None:
                dst.copy_(src)
        self.reset()
        return self

    def reorder(self, index):
        """Beam search: row r continues the caption that row index[r] held (narrator.py:223,337: `input_ids[beam_idx]`;
        the reference re-runs the prefixes, here the cached keys / values move with their captions). Only the filled
        positions are gathered; rows of one clip stay inside that clip, so the image keys / values are untouched."""
        n = self.steps
        if n == 0:
            return
        for c in self.cache:
            c[:, :n] = c[:, :n].index_select(0, index)

    def step(self, ids):
        if self.steps >= self.capacity:
            raise RuntimeError(f'DecodeSession: cache of {self.capacity} positions is full')
        self.ids.copy_(ids.reshape(-1))
        self.steps += 1
        if self._graph is not None:
            self._graph.replay()
            return self._logits
        return self._run()


class GPT2LMHeadModel(nn.Module):
    """gpt2_gated.py:1004-1162 (the language-model head on GPT2Model), inference only. `forward` takes the reference's
    keyword names; everything but input_ids / encoder_hidden_states / labels must be left at None."""

    def __init__(self, config):
        super().__init__()
        if _cfg(config, 'activation_function', 'gelu_new') != 'gelu_new':
            raise NotImplementedError('only GPT-2\'s gelu_new activation is built')
        if _cfg(config, 'scale_attn_by_inverse_layer_idx', False) or _cfg(config, 'reorder_and_upcast_attn', False) or \
                not _cfg(config, 'scale_attn_weights', True):
            raise NotImplementedError('attention variants other than GPT-2\'s 1/sqrt(d) scaling are not built')
        D, H = _cfg(config, 'n_embd'), _cfg(config, 'n_head')
        if D != H * 64:
            raise NotImplementedError(f'lavila_amd attention kernels are built for head_dim 64, got {D}/{H}')
        self.config = config
        self.transformer = GPT2Model(config)
        self.lm_head = nn.Linear(D, _cfg(config, 'vocab_size'), bias=False)
        if _cfg(config, 'tie_word_embeddings', True):
            self.lm_head.weight = self.transformer.wte.weight
        self._packs = {}
        self._sessions = {}

    def __getstate__(self):
        """The packed weights and decode sessions (device buffers, captured graphs) are caches of the parameters: they
        are rebuilt on demand and never copied or pickled with the module (copy.deepcopy, torch.save(model))."""
        state = self.__dict__.copy()
        state['_packs'], state['_sessions'] = {}, {}
        return state

    # ---- reference API ---------------------------------------------------------------------------------
    def freeze_lm_weights(self):
        """gpt2_gated.py:1019-1030."""
        for n, p in self.named_parameters():
            p.requires_grad = ('crossattention' in n or 'cross_attn' in n or 'alpha_cattn' in n or 'alpha_dense' in n)

    def gradient_checkpointing_enable(self):
        pass                                    # inference only: nothing is kept for a backward

    def gradient_checkpointing_disable(self):
        pass

    def get_output_embeddings(self):
        return self.lm_head

    def _param_key(self):
        """What the packed inference image was built from: the decoder's OWN parameters (version counter + storage) and a
        decoder-local generation. (Round 3 keyed on the process-wide weight-copy generation of ops.py, which every
        optimizer step and every grad-enabled model forward anywhere in the process bumps: a 1.5-B-parameter repack and a
        hipGraph recapture per batch for nothing -- ADVICE r3.) Writes that bypass the version counter (`param.data`
        edits, a fused optimizer stepping the decoder) need `invalidate_packed_weights()`."""
        return (getattr(self, '_pack_gen', 0),) + tuple((p._version, p.data_ptr()) for p in self.parameters())

    def invalidate_packed_weights(self):
        """Forget the packed bf16 image and the decode sessions built on it (they are rebuilt on the next call)."""
        self._pack_gen = getattr(self, '_pack_gen', 0) + 1
        self._packs, self._sessions = {}, {}

    clear_sessions = invalidate_packed_weights

    def _compute_dtype(self):
        lp = ops.autocast_dtype()
        if lp is not None:
            return lp
        dt = self.transformer.wte.weight.dtype
        return torch.bfloat16 if dt == torch.float16 else dt

    def _pack(self, dtype=None):
        dtype = dtype or self._compute_dtype()
        if dtype not in (torch.float32, torch.bfloat16):
            raise C.HipExtensionError(f'the decoder computes in float32 or bfloat16, not {dtype}')
        pack = self._packs.get(dtype)
        if pack is None or pack.key != self._param_key():
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                raise NotImplementedError('lavila_amd.gpt2_gated is inference-only (no backward kernels for the decoder): '
                                          'call it under torch.no_grad()')
            C.require_device(self.transformer.wte.weight)
            with torch.no_grad(), torch.autocast('cuda', enabled=False):
                pack = self._packs[dtype] = _Pack(self, dtype)
        return pack

    def forward(self, input_ids=None, past_key_values=None, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        unsupported = dict(past_key_values=past_key_values, attention_mask=attention_mask, token_type_ids=token_type_ids,
                           position_ids=position_ids, head_mask=head_mask, inputs_embeds=inputs_embeds,
                           encoder_attention_mask=encoder_attention_mask)
        bad = [k for k, v in unsupported.items() if v is not None]
        if bad or use_cache or output_attentions or output_hidden_states or return_dict is False:
            raise NotImplementedError(f'GPT2LMHeadModel.forward: {bad or "use_cache / output_* / return_dict=False"} is '
                                      'not on the narrator path (cached decoding: decode_session())')
        if input_ids is None:
            raise ValueError('You have to specify input_ids')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError('lavila_amd.gpt2_gated is inference-only (no backward kernels for the decoder): '
                                      'call it under torch.no_grad()')
        pack = self._pack()
        shape = tuple(input_ids.shape)
        L = shape[-1]
        if L > pack.positions:
            raise ValueError(f'sequence of {L} tokens exceeds the decoder\'s {pack.positions} positions')
        ids = input_ids.reshape(-1, L)
        B = ids.shape[0]
        lo, hi = (int(v) for v in torch.stack(torch.aminmax(ids)).tolist())      # ONE read-back; nn.Embedding would
        # fail on out-of-range ids, the gather kernel clamps
        if lo < 0 or hi >= pack.vocab:
            raise IndexError(f'input_ids out of range [0, {pack.vocab}): min {lo}, max {hi}')
        with torch.no_grad(), torch.autocast('cuda', enabled=False):
            xkv = qrep = None
            if encoder_hidden_states is not None and any(e['cross'] for e in pack.blocks):
                enc = ops.lowp(encoder_hidden_states)
                if enc.shape[0] != B or enc.shape[-1] != pack.D:
                    raise ValueError(f'encoder_hidden_states {tuple(enc.shape)} does not match {B} sequences of width {pack.D}')
                xkv, qrep = pack.image_kv(enc), L
            x = pack.embed(ids, L)

            def self_attention(i, qkv):
                return ops.causal_attention(qkv.reshape(B, L, 3 * pack.D), pack.heads).reshape(B * L, pack.D)
            h = pack.run(x, xkv, qrep, self_attention)
            logits = pack.logits(h).reshape(*shape, pack.vocab)
            loss = None
            if labels is not None:                              # gpt2_gated.py:1142-1148
                loss = F.cross_entropy(logits[..., :-1, :].reshape(-1, pack.vocab).float(), labels[..., 1:].reshape(-1))
        if self.transformer.wte.weight.dtype == torch.float16 and not torch.is_autocast_enabled():
            logits = logits.to(torch.float16)
        return CausalLMOutput(logits, loss)

    def decode_session(self, encoder_hidden_states, max_length, seqs_per_context=1, graph=True):
        """Cached decoding against `encoder_hidden_states` [contexts, NQ, D]; see DecodeSession. The last sessions are
        kept (buffers + captured graph) and re-bound when the next batch has the same shape -- the captioning drivers
        call generate() once per batch (main_infer_narrator.py:178-188) -- as long as the parameters have not changed."""
        pack = self._pack()
        enc = ops.lowp(encoder_hidden_states)
        key = (tuple(enc.shape), str(enc.device), int(max_length), int(seqs_per_context), bool(graph), pack.dtype)
        with torch.no_grad(), torch.autocast('cuda', enabled=False):
            sess = self._sessions.get(key)
            if sess is not None and sess.pack is pack:
                return sess.rebind(enc)
            # keep two: drop the older ones BEFORE the new session's buffers are allocated
            self._sessions = {k: v for k, v in list(self._sessions.items())[-1:] if v.pack is pack}
            sess = DecodeSession(self, pack, enc, max_length, seqs_per_context, graph)
            self._sessions[key] = sess
            return sess
