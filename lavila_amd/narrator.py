"""The narrator (BASELINE configs[4] / SURVEY.md section 8f rank 4), MI355X-native, inference only:
`VCLM_HF` = video tower (all-token features) -> attention pooling onto `num_img_queries` learned queries -> LayerNorm
(`encode_image`, narrator.py:63-87) -> gated-cross-attention GPT-2 (`lavila_amd.gpt2_gated`) -> `forward` (teacher-forced
logits, narrator.py:89-104) and `generate` (multinomial / top-k / top-p sampling with perplexities, narrator.py:106-147),
with the reference's module / parameter names (`visual.*`, `img_queries`, `img_attn_pool.{norm.gamma,
context_norm.gamma, to_q.weight, to_kv.weight, to_out.weight}`, `img_attn_pool_norm.gamma`, the `beta` buffers,
`text_decoder.*`) so that a reference `VCLM_*` checkpoint loads unchanged.

`CrossAttention` / `LayerNorm` mirror `lavila/models/coca.py:25-131` (same constructor); the pooling core is one
C-ABI call (lvl_mq_cross_attn_fwd), the projections go through ops.linear (own MFMA GEMMs where the widths tile),
the LayerNorms through lvl_layernorm_fwd. The pooling core has no backward kernel and says so.

`generate` keeps the reference's signature and bookkeeping (nll / entropy accumulation, eos tracking, teacher forcing,
num_return_sequences) but decodes against a key/value cache, one hipGraph replay per token (gpt2_gated.DecodeSession)
-- the reference re-runs the whole prefix for every token; `kv_cache=False` runs that schedule for comparison.
`beam_sample` / `group_beam_search` (narrator.py:149-366, built on transformers' BeamSearchScorer) are not built.
The drop-in package re-exports this module as `lavila.models.narrator` (and `lavila_amd.gpt2_gated` as
`lavila.models.gpt2_gated`, the two coca classes as `lavila.models.coca`).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

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


def sample_next_token(logits, top_k, top_p, temperature, target=None, pad_id=-100, uniform=None, debug=False):
    """One lvl_sample_next_token call on a step's logits [rows, vocab] (bf16, unit column stride, rows padded to a
    multiple of 8 columns -- what DecodeSession.step returns): -> (next_token [rows, 1] int64, nll [rows] f32,
    counted [rows] f32), or None when the tensor is not in that form (f32 logits of the parity configuration, the
    recompute schedule's strided slices): the caller then runs the framework ops. `uniform` [rows] in [0, 1) defaults to
    torch.rand on the device (torch's generator: torch.manual_seed reproduces a run)."""
    if logits.dtype != torch.bfloat16 or not logits.is_cuda or logits.dim() != 2 or logits.stride(1) != 1:
        return None
    rows, vocab = logits.shape
    stride = logits.stride(0)
    if stride % 8 != 0 or stride < ((vocab + 7) & ~7) or logits.data_ptr() % 16 != 0 or \
            vocab > C.lib().lvl_sample_max_vocab():
        return None
    dev = logits.device
    if uniform is None:
        uniform = torch.rand(rows, device=dev)
    tgt = None if target is None else target.contiguous()
    nxt = torch.empty(rows, dtype=torch.int64, device=dev)
    nll = torch.empty(rows, dtype=torch.float32, device=dev)
    cnt = torch.empty(rows, dtype=torch.float32, device=dev)
    dbg = torch.zeros(rows, 12, dtype=torch.float32, device=dev) if debug else None
    k = 0 if not top_k else min(int(top_k), vocab)
    p = 1.0 if top_p is None else float(top_p)
    t = 1.0 if temperature is None else float(temperature)
    if not (0.0 < p <= 1.0) or not (t > 0.0) or k < 0:
        return None          # outside the kernel's domain (the reference's warpers decide what happens): framework ops
    C.check(C.lib().lvl_sample_next_token(C.ptr(logits), stride, rows, vocab, t, k, p, C.ptr(uniform), C.ptr(tgt),
                                          int(pad_id) if pad_id is not None else -100, C.ptr(nxt), C.ptr(nll), C.ptr(cnt),
                                          C.ptr(dbg), C.stream_ptr()), 'lvl_sample_next_token')
    out = (nxt[:, None], nll, cnt)
    return out + (dbg,) if debug else out


class VCLM_HF(nn.Module):
    """narrator.py:31-147: same constructor; encode_image, forward and generate run on the HIP path (`text_decoder`:
    a lavila_amd.gpt2_gated.GPT2LMHeadModel), and so do beam_sample / group_beam_search (round 4)."""

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
            raise NotImplementedError('VCLM_HF.forward needs a text decoder (lavila_amd.gpt2_gated.GPT2LMHeadModel)')
        if use_checkpoint:
            self.text_decoder.gradient_checkpointing_enable()
        else:
            self.text_decoder.gradient_checkpointing_disable()
        text, labels = text[:, :-1], text[:, 1:]
        image_tokens = self.encode_image(image, use_checkpoint=use_checkpoint)
        logits = self.text_decoder(text.contiguous(), encoder_hidden_states=image_tokens).logits
        return {'text_tokens_logits': logits.permute(0, 2, 1), 'labels': labels}

    @staticmethod
    def _warp(logits, top_k, top_p, temperature):
        """The logits warpers narrator.py:368-389 builds for num_beams=1 (transformers' Temperature / TopK / TopP warpers,
        min_tokens_to_keep=1), restated: scale by 1/temperature; keep the top_k largest; drop the low-probability tail
        whose cumulative mass is <= 1 - top_p (always keeping the most probable token)."""
        if temperature is not None and temperature != 1.0:
            logits = logits / temperature
        if top_k is not None and top_k != 0:
            k = min(int(top_k), logits.shape[-1])
            kth = torch.topk(logits, k)[0][..., -1, None]
            logits = logits.masked_fill(logits < kth, float('-inf'))
        if top_p is not None and top_p < 1.0:
            sorted_logits, sorted_idx = torch.sort(logits, descending=False)
            cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
            remove = cum <= (1 - top_p)
            remove[..., -1:] = False
            logits = logits.masked_fill(remove.scatter(1, sorted_idx, remove), float('-inf'))
        return logits

    def generate(self, image_tokens, tokenizer, target=None, max_text_length=77, top_k=None, top_p=None,
                 num_return_sequences=1, temperature=1.0, teacher_forcing=False, early_stopping=False, kv_cache=True,
                 graph=True):
        """narrator.py:106-147, same arguments and return value (ids [B*n, <=max_text_length], perplexity [B*n]).
        Per step the decoder yields the next-token logits of every caption; the reference's bookkeeping follows on them
        unchanged (cross entropy against target[:, i+1] ignoring pad, or the entropy of the distribution while the row
        has not emitted eos; warp -> softmax -> multinomial). kv_cache=True (default): one new row per caption against
        cached keys / values, the image keys / values projected once per CLIP (shared by its num_return_sequences
        samples), one hipGraph replay per token when graph=True. kv_cache=False: the reference's schedule (the whole
        prefix through the decoder every step). bf16 logits go through ONE sampling kernel (lvl_sample_next_token:
        perplexity term, warpers and the draw; LAVILA_NARRATOR_SAMPLER=torch keeps the framework ops, which f32 logits
        always use): the same distribution as warp -> softmax -> multinomial, another mapping of random numbers to tokens."""
        n = int(num_return_sequences)
        B = image_tokens.shape[0] * n
        device = image_tokens.device
        bos, eos, pad = tokenizer.bos_token_id, tokenizer.eos_token_id, tokenizer.pad_token_id
        generated = torch.full((B, 1), bos, dtype=torch.long, device=device)
        condition = generated.clone()
        nlls = torch.zeros(B, device=device)
        num_tokens = torch.zeros(B, device=device)
        reached = torch.zeros(B, dtype=torch.bool, device=device)
        if kv_cache and teacher_forcing and target is not None and bool((target[:, 0] != bos).any()):
            kv_cache = False        # the reference conditions on target[:, :i+2] INCLUDING its first token (narrator.py:140)
        with torch.no_grad():
            session = repeated = None
            if kv_cache:
                session = self.text_decoder.decode_session(image_tokens, max_text_length, seqs_per_context=n, graph=graph)
            else:
                repeated = image_tokens.repeat_interleave(n, dim=0)
            fused = os.environ.get('LAVILA_NARRATOR_SAMPLER', 'fused') != 'torch'
            for i in range(max_text_length - 1):
                if kv_cache:
                    logits = session.step(condition[:, -1])
                else:
                    logits = self.text_decoder(condition.contiguous(), encoder_hidden_states=repeated).logits[:, -1, :]
                tgt = None if target is None else target[:, i + 1]
                drawn = sample_next_token(logits, top_k, top_p, temperature, tgt, pad) if fused else None
                if drawn is not None:                     # one kernel: perplexity term + warpers + draw
                    next_token, nll, counted = drawn
                    if target is not None:
                        nlls += nll
                        num_tokens += counted
                    else:
                        nlls += nll * (~reached)
                        num_tokens += (~reached)
                else:
                    logits = logits.float()
                    if target is not None:
                        nlls += F.cross_entropy(logits, tgt, ignore_index=pad, reduction='none')
                        num_tokens += tgt.ne(pad)
                    else:
                        nlls += torch.special.entr(F.softmax(logits, dim=1)).sum(dim=1) * (~reached)
                        num_tokens += (~reached)
                    if top_k == 1:                        # greedy: the first maximum (the reference's multinomial over a
                        next_token = logits.argmax(dim=-1, keepdim=True)     # one-hot; exact ties are not drawn among)
                    else:
                        probs = F.softmax(self._warp(logits, top_k, top_p, temperature), dim=-1)
                        next_token = torch.multinomial(probs, num_samples=1)
                reached = reached | (next_token[:, 0] == eos)
                if early_stopping and bool(torch.all(reached)):
                    break
                condition = target[:, :i + 2] if teacher_forcing else torch.cat((generated, next_token), dim=1)
                generated = torch.cat((generated, next_token), dim=1)
        return generated, torch.exp(nlls / num_tokens)

    # ---- beam search (narrator.py:149-366) ---------------------------------------------------------------------
    def _beam_logits(self, kv_cache, image_tokens, rows_per_clip, max_text_length, graph):
        """-> (step(input_ids) -> next-token logits [rows, vocab] float32, reorder(rows index)). kv_cache: one new row per
        beam against cached keys / values (the cache rows are re-gathered with the beams after every step); otherwise the
        reference's schedule: the whole prefix through the decoder every step (narrator.py:180-185,282-286)."""
        if kv_cache:
            session = self.text_decoder.decode_session(image_tokens, max_text_length, seqs_per_context=rows_per_clip,
                                                       graph=graph)
            return (lambda ids: session.step(ids[:, -1]).float()), session.reorder
        repeated = image_tokens.repeat_interleave(rows_per_clip, dim=0)
        return (lambda ids: self.text_decoder(ids.contiguous(), encoder_hidden_states=repeated).logits[:, -1, :].float(),
                lambda index: None)

    @staticmethod
    def _warp_beams(scores, top_k, top_p, temperature):
        """narrator.py:368-389 with num_beams > 1: the same warpers as _warp, each keeping at least TWO tokens per row."""
        if temperature is not None and temperature != 1.0:
            scores = scores / temperature
        if top_k is not None and top_k != 0:
            k = min(max(int(top_k), 2), scores.shape[-1])
            kth = torch.topk(scores, k)[0][..., -1, None]
            scores = scores.masked_fill(scores < kth, float('-inf'))
        if top_p is not None and top_p < 1.0:
            sorted_scores, sorted_idx = torch.sort(scores, descending=False)
            remove = sorted_scores.softmax(dim=-1).cumsum(dim=-1) <= (1 - top_p)
            remove[..., -2:] = False
            scores = scores.masked_fill(remove.scatter(1, sorted_idx, remove), float('-inf'))
        return scores

    def beam_sample(self, image_tokens, tokenizer, target=None, max_text_length=77, top_k=None, top_p=None,
                    temperature=1.0, length_penalty=1., num_beams=3, num_return_sequences=1, teacher_forcing=False,
                    early_stopping=False, kv_cache=True, graph=True):
        """narrator.py:149-241, same arguments and return value (sequences [B * num_return_sequences, <= max_text_length],
        sequence_scores): every clip runs num_return_sequences independent beam searches of num_beams beams; per step the
        2 * num_beams candidates of a search are DRAWN (multinomial without replacement over its beams' warped
        log-probabilities + beam scores, narrator.py:196-208), sorted and handed to the beam bookkeeping
        (lavila_amd.beam_search.BeamScorer = transformers' BeamSearchScorer). `target`, `teacher_forcing` and
        `early_stopping` are accepted and unused, as in the reference."""
        from .beam_search import BeamScorer
        if self.text_decoder is None:
            raise NotImplementedError('beam search needs a text decoder (lavila_amd.gpt2_gated.GPT2LMHeadModel)')
        batch = image_tokens.shape[0]
        device = image_tokens.device
        bos, eos, pad = tokenizer.bos_token_id, tokenizer.eos_token_id, tokenizer.pad_token_id
        per_clip = num_beams * num_return_sequences
        scorer = BeamScorer(batch * num_return_sequences, num_beams, device, length_penalty=length_penalty)
        entries = scorer.entries
        input_ids = torch.full((batch * per_clip, 1), bos, dtype=torch.long, device=device)
        beam_scores = torch.zeros(entries * num_beams, device=device)
        reached = torch.zeros(entries * num_beams, dtype=torch.bool, device=device)
        with torch.no_grad():
            step, reorder = self._beam_logits(kv_cache, image_tokens, per_clip, max_text_length, graph)
            for _ in range(max_text_length - 1):
                scores = F.log_softmax(step(input_ids), dim=-1) + beam_scores[:, None]
                scores = self._warp_beams(scores, top_k, top_p, temperature)
                vocab = scores.shape[-1]
                scores = scores.view(entries, num_beams * vocab)
                cand = torch.multinomial(F.softmax(scores, dim=-1), num_samples=2 * num_beams)
                cand_scores, order = torch.sort(torch.gather(scores, -1, cand), descending=True, dim=1)
                cand = torch.gather(cand, -1, order)
                beam_scores, new_tokens, rows = scorer.process(input_ids, cand_scores, cand % vocab,
                                                               torch.div(cand, vocab, rounding_mode='floor'), pad, eos)
                input_ids = torch.cat([input_ids[rows, :], new_tokens.unsqueeze(-1)], dim=-1)
                reorder(rows)
                reached = reached | (input_ids[:, -1] == eos)
                if scorer.is_done or bool(torch.all(reached)):
                    break
            return scorer.finalize(input_ids, beam_scores, max_text_length, pad, eos)

    def group_beam_search(self, image_tokens, tokenizer, target=None, max_text_length=77, top_k=None, top_p=None,
                          temperature=1.0, length_penalty=1., num_beams=6, num_beam_groups=3, num_return_sequences=1,
                          teacher_forcing=False, early_stopping=False, kv_cache=True, graph=True):
        """narrator.py:243-366, same arguments and return value: every clip runs num_beams beams in num_beam_groups groups;
        the groups take the top 2 * group_size candidates of their own beams in turn (no diversity penalty: the reference
        skips the logits processors, narrator.py:303-305) and the num_return_sequences best closed hypotheses of a clip
        are returned."""
        from .beam_search import BeamScorer
        if self.text_decoder is None:
            raise NotImplementedError('beam search needs a text decoder (lavila_amd.gpt2_gated.GPT2LMHeadModel)')
        batch = image_tokens.shape[0]
        device = image_tokens.device
        bos, eos, pad = tokenizer.bos_token_id, tokenizer.eos_token_id, tokenizer.pad_token_id
        scorer = BeamScorer(batch, num_beams, device, length_penalty=length_penalty, keep=num_return_sequences,
                            num_beam_groups=num_beam_groups)
        sub = num_beams // num_beam_groups
        input_ids = torch.full((batch * num_beams, 1), bos, dtype=torch.long, device=device)
        beam_scores = torch.full((batch, num_beams), -1e9, dtype=torch.float, device=device)
        beam_scores[:, ::sub] = 0
        beam_scores = beam_scores.view(batch * num_beams)
        reached = torch.zeros(batch * num_beams, dtype=torch.bool, device=device)
        base = torch.arange(batch, device=device)[:, None] * num_beams
        with torch.no_grad():
            step, reorder = self._beam_logits(kv_cache, image_tokens, num_beams, max_text_length, graph)
            current = torch.zeros(batch * num_beams, dtype=torch.long, device=device)
            gather_rows = torch.zeros(batch * num_beams, dtype=torch.long, device=device)
            for _ in range(max_text_length - 1):
                logits = step(input_ids)                                   # every beam of every group, once per step
                for gi in range(num_beam_groups):
                    lo = gi * sub
                    rows_g = (base + torch.arange(lo, lo + sub, device=device)[None, :]).reshape(-1)     # this group's rows
                    group_ids = input_ids[rows_g]
                    scores = F.log_softmax(logits[rows_g], dim=-1) + beam_scores[rows_g].unsqueeze(-1)
                    scores = self._warp_beams(scores, top_k, top_p, temperature)
                    vocab = scores.shape[-1]
                    cand_scores, cand = torch.topk(scores.view(batch, sub * vocab), 2 * sub, dim=1, largest=True,
                                                   sorted=True)
                    new_scores, new_tokens, picked = scorer.process(group_ids, cand_scores, cand % vocab,
                                                                    torch.div(cand, vocab, rounding_mode='floor'), pad, eos)
                    beam_scores[rows_g] = new_scores
                    input_ids[rows_g] = group_ids[picked]
                    current[rows_g] = new_tokens
                    # `picked` indexes the group's rows (entry * sub + beam): as rows of the whole batch
                    gather_rows[rows_g] = num_beams * torch.div(picked, sub, rounding_mode='floor') + lo + picked % sub
                input_ids = torch.cat([input_ids, current.unsqueeze(-1)], dim=-1)
                reorder(gather_rows)
                reached = reached | (input_ids[:, -1] == eos)
                if scorer.is_done or bool(torch.all(reached)):
                    break
            return scorer.finalize(input_ids, beam_scores, max_text_length, pad, eos)
