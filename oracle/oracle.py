"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32/fp64) restatement of the LaViLa dual-encoder
pretraining hot path. It is the checker for the HIP kernels; it is never the thing shipped or
measured (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Parity status: PINNED. The reference repository has no tests or golden vectors of its own
(SURVEY.md section 4), so the pins are outputs of the reference itself, generated in the build
container by oracle/gen_golden.py (which imports /root/reference unmodified) and committed under
tests/golden/. tests/test_oracle_golden.py checks every function here against them.

Every function cites the reference lines it restates (paths relative to the reference root).
The code is functional (explicit weight dicts with the reference's state_dict key names, explicit
index arithmetic instead of einops patterns) so that each step is one checkable formula.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# elementwise / normalisation
# --------------------------------------------------------------------------------------
def quick_gelu(x: Tensor) -> Tensor:
    """x * sigmoid(1.702 x)   -- lavila/models/openai_model.py:177-179"""
    return x * torch.sigmoid(1.702 * x)


def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    """Affine LayerNorm over the last dim, biased variance.
    eps = 1e-5 for ln_pre / text ln_1, ln_2, ln_final (timesformer.py:263-264, openai_model.py:187,193,
    models.py:106); eps = 1e-6 for norm1/2/3 and the final norm (timesformer.py:247)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * weight + bias


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    """F.normalize(dim=-1): x / max(||x||_2, eps)   -- lavila/models/models.py:168-170"""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


# --------------------------------------------------------------------------------------
# patch embedding + positional / temporal embedding
# --------------------------------------------------------------------------------------
def patchify(video_bcthw: Tensor, patch: int) -> Tensor:
    """[B,3,F,H,W] -> [B, F*N, 3*P*P] patch matrix; row order frame-major then (py,px); column
    order (c, i, j) -- the flattening of Conv2d weight [D,3,P,P].
    Restates permute(0,2,1,3,4) (timesformer.py:387) + Conv2d(k=stride=P) viewed as a GEMM
    (timesformer.py:77,79-84) + flatten(2).transpose(2,1).reshape(b,-1,D) (timesformer.py:349-350)."""
    B, C, Fr, H, W = video_bcthw.shape
    gh, gw = H // patch, W // patch
    x = video_bcthw.permute(0, 2, 1, 3, 4)                        # B F C H W
    x = x.reshape(B, Fr, C, gh, patch, gw, patch)
    x = x.permute(0, 1, 3, 5, 2, 4, 6)                            # B F gh gw C i j
    return x.reshape(B, Fr * gh * gw, C * patch * patch)


def patch_embed(video_bcthw: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """VideoPatchEmbed (timesformer.py:61-84) as patch-matrix x weight^T (+ bias iff ln_pre=False)."""
    D, C, P, _ = w.shape
    y = patchify(video_bcthw, P) @ w.reshape(D, C * P * P).t()
    return y if b is None else y + b


def total_pos_embed(pos_embed: Tensor, temporal_embed: Tensor, n_per_frame: int, frames: int) -> Tensor:
    """[1+frames*N, D]: row 0 = pos_embed[0]; row 1+f*N+n = pos_embed[1+n] + temporal_embed[f]
    -- timesformer.py:356-364 (tile over frames, repeat_interleave over patches, truncated to the
    current number of tokens)."""
    pos = pos_embed[0]                       # [N+1, D]
    tem = temporal_embed[0]                  # [num_frames, D]
    body = pos[1:].unsqueeze(0) + tem[:frames].unsqueeze(1)       # [F, N, D]
    return torch.cat([pos[:1], body.reshape(frames * n_per_frame, -1)], 0)


# --------------------------------------------------------------------------------------
# divided space-time attention (VarAttention)
# --------------------------------------------------------------------------------------
def _softmax_av(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """softmax(q k^T) v over the last two dims, no mask/dropout/extra scale -- timesformer.py:35-39"""
    s = q @ k.transpose(-1, -2)
    return torch.softmax(s, dim=-1) @ v


def divided_attention_core(qkv: Tensor, heads: int, frames: int, n_per_frame: int, mode: str) -> Tensor:
    """The part of VarAttention.forward between the qkv Linear and the proj Linear
    (timesformer.py:110-140), on packed qkv [B, T, 3*D] with T = 1 + frames*n_per_frame.

    * q is pre-scaled by dh^-0.5 (timesformer.py:113)
    * CLS query (token 0) attends to all T keys (timesformer.py:116-119)
    * patch queries are grouped: mode='space' -> per frame (N queries, keys = [cls] + N patches of
      that frame; 'b (f n) d -> (b f) n d', timesformer.py:300-301); mode='time' -> per location
      (F queries, keys = [cls] + F patches of that location; '(b n) f d', timesformer.py:302-303)
    * output [B, T, D] with heads merged '(b h) n d -> b n (h d)' (timesformer.py:134-140)
    """
    B, T, D3 = qkv.shape
    D = D3 // 3
    dh = D // heads
    Fr, N = frames, n_per_frame
    assert T == 1 + Fr * N
    q, k, v = qkv.reshape(B, T, 3, heads, dh).permute(2, 0, 3, 1, 4)      # each [B,H,T,dh]
    q = q * (dh ** -0.5)
    out = torch.empty(B, heads, T, dh, dtype=qkv.dtype)
    out[:, :, 0:1] = _softmax_av(q[:, :, 0:1], k, v)
    qp = q[:, :, 1:].reshape(B, heads, Fr, N, dh)
    kp = k[:, :, 1:].reshape(B, heads, Fr, N, dh)
    vp = v[:, :, 1:].reshape(B, heads, Fr, N, dh)
    if mode == 'time':                                                      # groups = locations
        qp, kp, vp = (t.transpose(2, 3) for t in (qp, kp, vp))             # [B,H,N,F,dh]
    G = qp.shape[2]
    kc = k[:, :, None, 0:1].expand(B, heads, G, 1, dh)
    vc = v[:, :, None, 0:1].expand(B, heads, G, 1, dh)
    og = _softmax_av(qp, torch.cat([kc, kp], 3), torch.cat([vc, vp], 3))
    if mode == 'time':
        og = og.transpose(2, 3)
    out[:, :, 1:] = og.reshape(B, heads, Fr * N, dh)
    return out.permute(0, 2, 1, 3).reshape(B, T, D)


def var_attention(x: Tensor, w: Dict[str, Tensor], prefix: str, heads: int, frames: int,
                  n_per_frame: int, mode: str) -> Tensor:
    """VarAttention.forward (timesformer.py:107-144): qkv Linear -> core -> proj Linear (dropouts p=0)."""
    qkv = F.linear(x, w[prefix + 'qkv.weight'], w.get(prefix + 'qkv.bias'))
    o = divided_attention_core(qkv, heads, frames, n_per_frame, mode)
    return F.linear(o, w[prefix + 'proj.weight'], w[prefix + 'proj.bias'])


def space_time_block(x: Tensor, w: Dict[str, Tensor], prefix: str, heads: int, frames: int,
                     n_per_frame: int, eps: float = 1e-6) -> Tensor:
    """SpaceTimeBlock.forward, 'frozen-in-time' wiring (timesformer.py:173-198):
        t  = x + [tanh(alpha)*] timeattn(norm3(x))
        x1 = x + attn(norm1(t))              <- residual from x, NOT from t
        x2 = x1 + mlp(norm2(x1)),  mlp = fc2(QuickGELU(fc1(.)))   (timesformer.py:52-58)"""
    p = prefix
    t_out = var_attention(layer_norm(x, w[p + 'norm3.weight'], w[p + 'norm3.bias'], eps),
                          w, p + 'timeattn.', heads, frames, n_per_frame, 'time')
    if p + 'alpha_timeattn' in w:
        t_out = torch.tanh(w[p + 'alpha_timeattn']) * t_out
    t = x + t_out
    s_out = var_attention(layer_norm(t, w[p + 'norm1.weight'], w[p + 'norm1.bias'], eps),
                          w, p + 'attn.', heads, frames, n_per_frame, 'space')
    x1 = x + s_out
    h = layer_norm(x1, w[p + 'norm2.weight'], w[p + 'norm2.bias'], eps)
    h = quick_gelu(F.linear(h, w[p + 'mlp.fc1.weight'], w[p + 'mlp.fc1.bias']))
    return x1 + F.linear(h, w[p + 'mlp.fc2.weight'], w[p + 'mlp.fc2.bias'])


def vision_tower(video_bcthw: Tensor, w: Dict[str, Tensor], heads: int, prefix: str = 'visual.',
                 cls_at_last: bool = True) -> Tensor:
    """SpaceTimeTransformer.forward/forward_features (timesformer.py:345-390) for the CLIP_OPENAI_*
    configuration (ln_pre=True, QuickGELU, head/pre_logits = Identity, models.py:347-349)."""
    p = prefix
    B, C, Fr, H, W = video_bcthw.shape
    pw = w[p + 'patch_embed.proj.weight']
    P = pw.shape[-1]
    N = (H // P) * (W // P)
    depth = 1 + max(int(k[len(p) + 7:].split('.')[0]) for k in w if k.startswith(p + 'blocks.'))
    x = patch_embed(video_bcthw, pw, w.get(p + 'patch_embed.proj.bias'))
    x = torch.cat([w[p + 'cls_token'].expand(B, -1, -1), x], 1)
    x = x + total_pos_embed(w[p + 'pos_embed'], w[p + 'temporal_embed'], N, Fr)
    if p + 'ln_pre.weight' in w:
        x = layer_norm(x, w[p + 'ln_pre.weight'], w[p + 'ln_pre.bias'], 1e-5)
    for i in range(depth):
        x = space_time_block(x, w, f'{p}blocks.{i}.', heads, Fr, N)
    x = layer_norm(x, w[p + 'norm.weight'], w[p + 'norm.bias'], 1e-6)
    return x[:, 0] if cls_at_last else x


# --------------------------------------------------------------------------------------
# text tower (OpenAI-CLIP Transformer)
# --------------------------------------------------------------------------------------
def cls_attention_core(q: Tensor, kv: Tensor, heads: int) -> Tensor:
    """The cls row of VarAttention's attention (timesformer.py:113-119): q [B, D] of the cls token (times dh^-0.5) against
    the keys / values kv [B, T, 2D] of ALL tokens, per head."""
    B, T, D2 = kv.shape
    D = D2 // 2
    qh = q.reshape(B, heads, 1, 64) * (64 ** -0.5)
    k = kv[..., :D].reshape(B, T, heads, 64).permute(0, 2, 1, 3)
    v = kv[..., D:].reshape(B, T, heads, 64).permute(0, 2, 1, 3)
    return _softmax_av(qh, k, v).reshape(B, D)


def coca_layer_norm(x: Tensor, gamma: Tensor) -> Tensor:
    """coca.py:27-34: F.layer_norm with a learned gamma and a zero beta buffer (eps 1e-5)."""
    return F.layer_norm(x, x.shape[-1:], gamma, torch.zeros_like(gamma), 1e-5)


def mq_cross_attention_core(q: Tensor, kv: Tensor, heads: int) -> Tensor:
    """coca.py:104-120 between to_q / to_kv and to_out: q [B,n,heads*64] (times dim_head^-0.5), ONE key/value head
    kv [B,j,128] = k | v shared by all query heads, max-subtracted softmax over j."""
    B, n, _ = q.shape
    qh = q.reshape(B, n, heads, 64).permute(0, 2, 1, 3) * (64 ** -0.5)
    k, v = kv.chunk(2, dim=-1)
    sim = torch.einsum('bhid,bjd->bhij', qh, k)
    attn = (sim - sim.amax(dim=-1, keepdim=True)).softmax(dim=-1)
    out = torch.einsum('bhij,bjd->bhid', attn, v)
    return out.permute(0, 2, 1, 3).reshape(B, n, heads * 64)


def cross_attention_pool(x: Tensor, context: Tensor, w: Dict[str, Tensor], prefix: str, heads: int) -> Tensor:
    """coca.CrossAttention.forward with norm_context=True, parallel_ff=False (coca.py:93-131)."""
    xq = coca_layer_norm(x, w[prefix + 'norm.gamma'])
    ctx = coca_layer_norm(context, w[prefix + 'context_norm.gamma'])
    q = F.linear(xq, w[prefix + 'to_q.weight'])
    kv = F.linear(ctx, w[prefix + 'to_kv.weight'])
    return F.linear(mq_cross_attention_core(q, kv, heads), w[prefix + 'to_out.weight'])


def narrator_encode_image(video_bcthw: Tensor, w: Dict[str, Tensor], vis_heads: int, pool_heads: int) -> Tensor:
    """VCLM_HF.encode_image (narrator.py:63-90): all-token tower features -> attention pooling onto img_queries ->
    img_attn_pool_norm."""
    feats = vision_tower(video_bcthw, w, vis_heads, cls_at_last=False)
    B = feats.shape[0]
    q = w['img_queries'][None].expand(B, -1, -1)
    pooled = cross_attention_pool(q, feats, w, 'img_attn_pool.', pool_heads)
    return coca_layer_norm(pooled, w['img_attn_pool_norm.gamma'])


# --------------------------------------------------------------------------------------
# narrator decoder (gated-cross-attention GPT-2) and greedy / teacher-forced decoding
# --------------------------------------------------------------------------------------
def gelu_new(x: Tensor) -> Tensor:
    """GPT-2's tanh GELU (config.activation_function == 'gelu_new', gpt2_gated.py:389 via transformers ACT2FN):
    0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def sq_relu(x: Tensor) -> Tensor:
    """relu(x)^2 of the cross-attention MLP (gpt2_gated.py:363-376, 413)."""
    return torch.relu(x) ** 2


def conv1d(x: Tensor, weight_in_out: Tensor, bias: Tensor) -> Tensor:
    """transformers Conv1D (pytorch_utils.py, used at gpt2_gated.py:184-188,383-384): y = x @ W + b with W stored
    [in, out] -- the TRANSPOSE of nn.Linear's layout."""
    return x @ weight_in_out + bias


def gpt2_attention_core(q: Tensor, k: Tensor, v: Tensor, heads: int, causal: bool) -> Tensor:
    """GPT2Attention._attn + head split / merge (gpt2_gated.py:206-307) for the configuration the narrator uses
    (scale_attn_weights, no layer-index scaling, no reordering): softmax(q k^T / sqrt(dh) [causal: where(tril, ., -1e4)]) v.
    q [B, Lq, D], k / v [B, Lk, D]; causal rows are the LAST Lq positions of the Lk keys (`bias[Lk-Lq:Lk, :Lk]`)."""
    B, Lq, D = q.shape
    Lk = k.shape[1]
    dh = D // heads
    qh = q.reshape(B, Lq, heads, dh).permute(0, 2, 1, 3)
    kh = k.reshape(B, Lk, heads, dh).permute(0, 2, 1, 3)
    vh = v.reshape(B, Lk, heads, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / (dh ** 0.5)
    if causal:
        allowed = torch.ones(Lk, Lk, dtype=torch.bool).tril_()[Lk - Lq:Lk, :Lk]
        s = torch.where(allowed, s, torch.tensor(-1e4, dtype=s.dtype))
    o = torch.softmax(s, -1) @ vh
    return o.permute(0, 2, 1, 3).reshape(B, Lq, D)


def gpt2_mlp(x: Tensor, w: Dict[str, Tensor], prefix: str, squared_relu: bool) -> Tensor:
    """GPT2MLP.forward (gpt2_gated.py:391-396); dropout is identity in eval."""
    h = conv1d(x, w[prefix + 'c_fc.weight'], w[prefix + 'c_fc.bias'])
    h = sq_relu(h) if squared_relu else gelu_new(h)
    return conv1d(h, w[prefix + 'c_proj.weight'], w[prefix + 'c_proj.bias'])


def gpt2_block(x: Tensor, enc: Optional[Tensor], w: Dict[str, Tensor], prefix: str, heads: int, eps: float,
               past: Optional[tuple] = None):
    """GPT2Block.forward (gpt2_gated.py:421-495). When the block owns a `crossattention` (layer_idx % freq == 0) and
    encoder states are given: x += tanh(alpha_cattn) * CrossAttn(ln_cross_attn x, enc); x += tanh(alpha_dense) *
    MLP_sqrelu(ln_2_crossattention x) (gates only when the parameters exist), THEN the ordinary GPT-2 block
    (causal self-attention, gelu_new MLP). Returns (x, (k, v)): the self-attention keys / values incl. `past`."""
    p = prefix
    D = x.shape[-1]
    if enc is not None and (p + 'crossattention.q_attn.weight') in w:
        h = layer_norm(x, w[p + 'ln_cross_attn.weight'], w[p + 'ln_cross_attn.bias'], eps)
        q = conv1d(h, w[p + 'crossattention.q_attn.weight'], w[p + 'crossattention.q_attn.bias'])
        kv = conv1d(enc, w[p + 'crossattention.c_attn.weight'], w[p + 'crossattention.c_attn.bias'])
        a = gpt2_attention_core(q, kv[..., :D], kv[..., D:], heads, causal=False)
        a = conv1d(a, w[p + 'crossattention.c_proj.weight'], w[p + 'crossattention.c_proj.bias'])
        if (p + 'alpha_cattn') in w:
            a = torch.tanh(w[p + 'alpha_cattn']) * a
        x = x + a
        h = layer_norm(x, w[p + 'ln_2_crossattention.weight'], w[p + 'ln_2_crossattention.bias'], eps)
        f = gpt2_mlp(h, w, p + 'mlp_crossattention.', squared_relu=True)
        if (p + 'alpha_dense') in w:
            f = torch.tanh(w[p + 'alpha_dense']) * f
        x = x + f
    h = layer_norm(x, w[p + 'ln_1.weight'], w[p + 'ln_1.bias'], eps)
    qkv = conv1d(h, w[p + 'attn.c_attn.weight'], w[p + 'attn.c_attn.bias'])
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    if past is not None:
        k = torch.cat([past[0], k], 1)
        v = torch.cat([past[1], v], 1)
    a = gpt2_attention_core(q, k, v, heads, causal=True)
    x = x + conv1d(a, w[p + 'attn.c_proj.weight'], w[p + 'attn.c_proj.bias'])
    h = layer_norm(x, w[p + 'ln_2.weight'], w[p + 'ln_2.bias'], eps)
    return x + gpt2_mlp(h, w, p + 'mlp.', squared_relu=False), (k, v)


def gpt2_lm_logits(ids: Tensor, enc: Optional[Tensor], w: Dict[str, Tensor], heads: int, eps: float = 1e-5,
                   prefix: str = '', past: Optional[list] = None):
    """GPT2LMHeadModel.forward -> logits (gpt2_gated.py:802-1001, 1092-1162): wte[ids] + wpe[past_len + arange(L)] ->
    blocks -> ln_f -> lm_head (no bias). `past`: per-layer (k, v) of the tokens before `ids` (None = none); returns
    (logits [B, L, V], presents)."""
    p = prefix + 'transformer.'
    depth = 1 + max(int(k[len(p) + 2:].split('.')[0]) for k in w if k.startswith(p + 'h.'))
    past_len = 0 if past is None else past[0][0].shape[1]
    L = ids.shape[1]
    x = w[p + 'wte.weight'][ids] + w[p + 'wpe.weight'][past_len:past_len + L]
    presents = []
    for i in range(depth):
        x, kv = gpt2_block(x, enc, w, f'{p}h.{i}.', heads, eps, None if past is None else past[i])
        presents.append(kv)
    x = layer_norm(x, w[p + 'ln_f.weight'], w[p + 'ln_f.bias'], eps)
    return x @ w[prefix + 'lm_head.weight'].t(), presents


def narrator_forward(video_bcthw: Tensor, text: Tensor, w: Dict[str, Tensor], vis_heads: int, pool_heads: int,
                     dec_heads: int) -> Dict[str, Tensor]:
    """VCLM_HF.forward (narrator.py:89-104): decoder over text[:, :-1] against the pooled image tokens; logits come
    back class-major [B, V, L-1] with labels text[:, 1:]."""
    image_tokens = narrator_encode_image(video_bcthw, w, vis_heads, pool_heads)
    logits, _ = gpt2_lm_logits(text[:, :-1], image_tokens, w, dec_heads, prefix='text_decoder.')
    return {'text_tokens_logits': logits.permute(0, 2, 1), 'labels': text[:, 1:]}


def narrator_generate_greedy(image_tokens: Tensor, w: Dict[str, Tensor], dec_heads: int, bos: int, eos: int, pad: int,
                             max_text_length: int, target: Optional[Tensor] = None, teacher_forcing: bool = False,
                             early_stopping: bool = False, use_cache: bool = True):
    """VCLM_HF.generate (narrator.py:106-147) with top_k=1, i.e. the multinomial draw is an argmax (deterministic): the
    only sampling setting that has a reference answer. Per step: logits of the last position -> nll (cross entropy
    against target[:, i+1] ignoring pad, or the entropy of the softmax while the row has not emitted eos) -> next token.
    `use_cache=False` re-runs the whole prefix each step exactly like the reference; `use_cache=True` feeds one token
    against cached keys / values (same numbers up to rounding). Returns (ids [B, <=max_text_length], perplexity [B])."""
    B = image_tokens.shape[0]
    generated = torch.full((B, 1), bos, dtype=torch.long)
    condition = generated.clone()
    nlls = torch.zeros(B)
    num = torch.zeros(B)
    reached = torch.zeros(B, dtype=torch.bool)
    past = None
    for i in range(max_text_length - 1):
        if use_cache:
            logits, past = gpt2_lm_logits(condition[:, -1:], image_tokens, w, dec_heads, prefix='text_decoder.', past=past)
        else:
            logits, _ = gpt2_lm_logits(condition, image_tokens, w, dec_heads, prefix='text_decoder.')
        nxt = logits[:, -1, :]
        if target is not None:
            nlls += F.cross_entropy(nxt, target[:, i + 1], ignore_index=pad, reduction='none')
            num += target[:, i + 1].ne(pad)
        else:
            nlls += torch.special.entr(F.softmax(nxt, dim=1)).sum(dim=1) * (~reached)
            num += (~reached)
        tok = nxt.argmax(-1, keepdim=True)
        reached = reached | (tok[:, 0] == eos)
        if early_stopping and bool(torch.all(reached)):
            break
        condition = target[:, :i + 2] if teacher_forcing else torch.cat([generated, tok], 1)
        generated = torch.cat([generated, tok], 1)
    return generated, torch.exp(nlls / num)


def causal_attention_core(qkv: Tensor, heads: int) -> Tensor:
    """nn.MultiheadAttention core with the additive causal mask of CLIP.build_attention_mask
    (models.py:131-137; openai_model.py:196-198): softmax((q dh^-0.5) k^T + mask) v, packed
    qkv [B, L, 3*W] -> [B, L, W]."""
    B, L, W3 = qkv.shape
    W = W3 // 3
    dh = W // heads
    q, k, v = qkv.reshape(B, L, 3, heads, dh).permute(2, 0, 3, 1, 4)
    s = (q * dh ** -0.5) @ k.transpose(-1, -2)
    mask = torch.full((L, L), float('-inf'), dtype=qkv.dtype).triu_(1)
    o = torch.softmax(s + mask, -1) @ v
    return o.permute(0, 2, 1, 3).reshape(B, L, W)


def text_block(x: Tensor, w: Dict[str, Tensor], prefix: str, heads: int) -> Tensor:
    """ResidualAttentionBlock.forward (openai_model.py:182-216): x + MHA(ln_1 x); x + MLP(ln_2 x)."""
    p = prefix
    h = layer_norm(x, w[p + 'ln_1.weight'], w[p + 'ln_1.bias'], 1e-5)
    qkv = F.linear(h, w[p + 'attn.in_proj_weight'], w[p + 'attn.in_proj_bias'])
    o = causal_attention_core(qkv, heads)
    x = x + F.linear(o, w[p + 'attn.out_proj.weight'], w[p + 'attn.out_proj.bias'])
    h = layer_norm(x, w[p + 'ln_2.weight'], w[p + 'ln_2.bias'], 1e-5)
    h = quick_gelu(F.linear(h, w[p + 'mlp.c_fc.weight'], w[p + 'mlp.c_fc.bias']))
    return x + F.linear(h, w[p + 'mlp.c_proj.weight'], w[p + 'mlp.c_proj.bias'])


def text_tower(tokens: Tensor, w: Dict[str, Tensor], heads: int) -> Tensor:
    """CLIP.encode_text (models.py:150-162): embed + pos -> blocks -> ln_final -> row at argmax(token
    id) (EOT = highest id) -> @ text_projection."""
    x = w['token_embedding.weight'][tokens] + w['positional_embedding']
    depth = 1 + max(int(k.split('.')[2]) for k in w if k.startswith('transformer.resblocks.'))
    for i in range(depth):
        x = text_block(x, w, f'transformer.resblocks.{i}.', heads)
    x = layer_norm(x, w['ln_final.weight'], w['ln_final.bias'], 1e-5)
    eot = tokens.argmax(-1)
    return x[torch.arange(x.shape[0]), eot] @ w['text_projection']


# --------------------------------------------------------------------------------------
# dual encoder + contrastive loss
# --------------------------------------------------------------------------------------
def clip_forward(video: Tensor, tokens: Tensor, w: Dict[str, Tensor], vision_heads: int,
                 text_heads: int, norm_embed: bool = False) -> Dict[str, Tensor]:
    """CLIP.forward (models.py:164-173) -> {'image_embed','text_embed','logit_scale'=exp(param)}"""
    img = vision_tower(video, w, vision_heads) @ w['image_projection']     # models.py:139-148
    txt = text_tower(tokens, w, text_heads)
    if norm_embed:
        img, txt = l2_normalize(img), l2_normalize(txt)
    return {'image_embed': img, 'text_embed': txt, 'logit_scale': w['logit_scale'].exp()}


def clip_logits(all_img: Tensor, all_txt: Tensor, logit_scale: Tensor) -> Tensor:
    """logits_per_image = logit_scale * all_img @ all_txt.T  (left-to-right) -- loss.py:78"""
    return (logit_scale * all_img) @ all_txt.t()


def clip_loss(all_img: Tensor, all_txt: Tensor, logit_scale: Tensor) -> Dict[str, Tensor]:
    """CLIPLoss.forward on the rank-ordered concatenation of all ranks' embeddings
    (loss.py:69-118; world_size>1 + use_vissl gathers in rank order, distributed_utils.py:88).
    labels = arange(G) int64; loss = (CE(Li)+CE(Li^T))/2; acc = 100*mean(argmax(Li,-1)==labels)."""
    li = clip_logits(all_img, all_txt, logit_scale)
    G = li.shape[0]
    labels = torch.arange(G, dtype=torch.long)
    loss = (F.cross_entropy(li, labels) + F.cross_entropy(li.t(), labels)) / 2
    pred = li.argmax(-1)
    acc = 100.0 * (pred == labels).sum() / G
    return {'loss': loss, 'clip_loss': loss, 'clip_acc': acc, 'logits_per_image': li,
            'labels': labels, 'pred': pred}


def ssl_scale_matrix(ind: Tensor, logit_scale: Tensor, logit_scale_pseudo: Tensor) -> Tensor:
    """Per-pair temperature of SSLCLIPLoss (loss.py:160-166 / 172-178): with mask[i][j] = ind[i] + ind[j]
    (gt_indicators: 1 = ground-truth narration, 0 = pseudo-label), scale = pseudo (mask 0),
    sqrt(pseudo * real) (mask 1), real (mask 2)."""
    mask = ind[:, None] + ind[None, :]
    geo = torch.sqrt(logit_scale_pseudo * logit_scale)
    one = torch.ones((), dtype=logit_scale.dtype)
    return torch.where(mask == 0, logit_scale_pseudo * one, torch.where(mask == 1, geo * one, logit_scale * one))


def ssl_clip_loss(all_img: Tensor, all_txt: Tensor, ind: Tensor, logit_scale: Tensor,
                  logit_scale_pseudo: Tensor) -> Dict[str, Tensor]:
    """SSLCLIPLoss.forward on the rank-ordered concatenation (loss.py:146-217): logits =
    scale_matrix * (img @ txt^T), symmetric InfoNCE, accuracy split by indicator."""
    li = ssl_scale_matrix(ind, logit_scale, logit_scale_pseudo) * (all_img @ all_txt.t())
    G = li.shape[0]
    labels = torch.arange(G, dtype=torch.long)
    loss = (F.cross_entropy(li, labels) + F.cross_entropy(li.t(), labels)) / 2
    pred = li.argmax(-1)
    ok = pred == labels
    gt, ps = ind == 1, ind == 0
    return {'loss': loss, 'clip_loss': loss, 'clip_acc': 100.0 * ok.sum() / G,
            'clip_acc_gt': 100.0 * ok[gt].sum() / gt.sum(), 'clip_acc_pseudo': 100.0 * ok[ps].sum() / ps.sum(),
            'num_gt': gt.sum(), 'num_pseudo': ps.sum(), 'logits_per_image': li, 'pred': pred, 'labels': labels}


def ssl_synthetic_inputs(G: int, E: int, seed: int):
    """Seeded SSLCLIPLoss inputs shared by the golden generator and the tests: unit-norm image/text rows
    (text correlated with its image), indicators ~Bernoulli(0.5) with both kinds forced present."""
    g = torch.Generator().manual_seed(seed)
    img = l2_normalize(torch.randn(G, E, generator=g))
    txt = l2_normalize(torch.randn(G, E, generator=g) + 0.7 * img)
    ind = (torch.rand(G, generator=g) < 0.5).long()
    ind[0], ind[1] = 1, 0
    return img, txt, ind


# --------------------------------------------------------------------------------------
# deterministic synthetic weights / inputs shared by the golden generator, tests and bench
# --------------------------------------------------------------------------------------
def synthetic_batch(batch: int, frames: int, img: int, seed: int = 1234, real_tokens: int = 32,
                    ctx: int = 77, spread: bool = False):
    """randn frames [B,3,F,H,W]; tokens [B,77] = SOT, 30 random ids, EOT at 31,
    zero padding (EOT=49407 is the max id so argmax finds it, tokenizer.py:147-162).

    spread=True (round 5 fixtures): the same noise plus what makes the SAMPLES differ after the towers' pooling -- iid
    noise alone averages out over a clip's patches and every clip lands on the same embedding (cosines 0.99 between
    samples, one argmax for the whole batch): a per-(sample, channel, frame) offset, a smooth per-sample pattern, and
    captions of ragged lengths (EOT anywhere in 5..real_tokens, so the EOT row, the causal trim and the padding differ
    per sample)."""
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(batch, 3, frames, img, img, generator=g)
    tokens = torch.zeros(batch, ctx, dtype=torch.long)
    tokens[:, 0] = 49406
    if not spread:
        tokens[:, 1:real_tokens - 1] = torch.randint(1, 49406, (batch, real_tokens - 2), generator=g)
        tokens[:, real_tokens - 1] = 49407
        return video, tokens
    video = video + 2.0 * torch.randn(batch, 3, frames, 1, 1, generator=g)
    yy = torch.linspace(0, 1, img).view(1, 1, 1, -1, 1)
    xx = torch.linspace(0, 1, img).view(1, 1, 1, 1, -1)
    k = (torch.arange(batch).float() % 7 + 1).view(batch, 1, 1, 1, 1)
    ph = torch.rand(batch, 3, 1, 1, 1, generator=g) * 6.2831853
    video = video + 2.0 * torch.sin(k * 6.2831853 * xx + ph) * torch.cos((8 - k) * 3.1415927 * yy + ph.roll(1, 1))
    lens = torch.randint(5, real_tokens + 1, (batch,), generator=g)
    lens[0] = real_tokens                           # the longest caption is in the batch
    for b in range(batch):
        n = int(lens[b])
        tokens[b, 1:n - 1] = torch.randint(1, 49406, (n - 2,), generator=g)
        tokens[b, n - 1] = 49407
    return video, tokens


def procedural_weights(shapes: Dict[str, tuple], seed: int = 0, spread: bool = False) -> Dict[str, Tensor]:
    """Deterministic weights from (name -> shape): every tensor gets its own CPU generator seeded by
    (seed, index in sorted-name order) so that the golden script, the tests and bench.py can rebuild
    identical weights without shipping them. Scales are chosen so the temporal path is live
    (the shipped zeros-init makes temporal attention a no-op).

    spread=True (round 5 fixtures): token embeddings of std 0.1, 2x larger in_proj weights in the text tower and 1.5x
    larger qkv weights in the video tower -- sharper attention than the near-uniform one of a randomly initialised deep
    transformer, which collapses every sample onto one embedding. With synthetic_batch(spread=True): mean cosine between
    samples 0.5 (video) / 0.4 (text) instead of 0.99, distinct argmax per row, and a softmax that is no longer flat.
    How far this can be pushed is set by CONDITIONING, measured before choosing: at 3x the video tower is past the edge
    of chaos (float32 and float64 evaluations of the SAME network differ by 5e-2); at 2x / 3x (video / text) float32
    evaluations agree to 1e-5 but the network amplifies a 2^-17 operand perturbation -- the f32-class mode of the MFMA
    kernels, 16-bit mantissa images -- to 3e-3 on a logit (measured on the GPU; CPU emulation of the operand rounding:
    7e-4 from the Linear layers alone), beyond north_star's 1e-3; at 1.5x / 2x the same emulation gives 9e-5."""
    out = {}
    for idx, name in enumerate(sorted(shapes)):
        shape = tuple(shapes[name])
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        leaf = name.split('.')[-1]
        if name == 'logit_scale':
            t = torch.tensor(math.log(1 / 0.07))
        elif name.endswith('alpha_timeattn') or leaf in ('alpha_cattn', 'alpha_dense'):
            t = torch.randn(shape, generator=g) * 0.5
        elif name.endswith('transformer.ln_f.weight'):      # GPT-2 final LayerNorm: logits of a few units
            t = 4.0 + 0.4 * torch.randn(shape, generator=g)
        elif leaf == 'weight' and len(shape) == 1:          # LayerNorm gains
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf in ('bias', 'in_proj_bias'):
            t = 0.02 * torch.randn(shape, generator=g)
        elif name in ('image_projection', 'text_projection'):
            t = torch.randn(shape, generator=g) * shape[0] ** -0.5
        elif leaf in ('cls_token', 'pos_embed', 'temporal_embed', 'positional_embedding'):
            t = 0.02 * torch.randn(shape, generator=g)
        elif name == 'token_embedding.weight':
            t = (0.1 if spread else 0.02) * torch.randn(shape, generator=g)
        elif 'patch_embed' in name:
            t = torch.randn(shape, generator=g) * (shape[1] * shape[2] * shape[3]) ** -0.5
        else:                                               # Linear / in_proj weights [out, in]
            t = torch.randn(shape, generator=g) * shape[-1] ** -0.5
            if spread and leaf == 'in_proj_weight':
                t = t * 2.0
            elif spread and name.endswith('qkv.weight'):
                t = t * 1.5
        out[name] = t
    return out


def narrator_weights(shapes: Dict[str, tuple], seed: int) -> Dict[str, Tensor]:
    """procedural_weights for a VCLM_HF state dict (tests/golden/narrator_decoder.pt lists `shapes` without the causal-mask
    buffers and without lm_head): the coca `beta` buffers are zeros (coca.py:31) and lm_head is tied to wte
    (GPT-2's tie_word_embeddings)."""
    w = procedural_weights({k: v for k, v in shapes.items() if not k.endswith('.beta')}, seed=seed)
    for k, shape in shapes.items():
        if k.endswith('.beta'):
            w[k] = torch.zeros(shape)
    w['text_decoder.lm_head.weight'] = w['text_decoder.transformer.wte.weight']
    return w
