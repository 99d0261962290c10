/*
 * lavila_hip.h -- C ABI of liblavila_hip.so: the MI355X (gfx950) hot kernels of the LaViLa
 * dual-encoder pretraining path.
 *
 * The reference (facebookresearch/LaViLa) has no native layer: every entry point below replaces a
 * span of stock torch ops inside a reference Python function (cited per function, paths relative
 * to the reference root). The host-side mirror of the reference API (the lavila_amd python package) binds these
 * through ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes; all pointers are DEVICE pointers (HBM) unless stated otherwise
 *   - `dtype` selects the activation element type: LVL_F32 (parity path) or LVL_BF16 (perf path);
 *     statistics, parameters (gamma/beta/bias/pos-embeds) and workspaces are always float32
 *   - row-major, innermost dimension contiguous, base pointers 16-byte aligned
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous and stream-ordered,
 *     never allocates, never synchronises, and is hipGraph-capturable
 *   - return value: 0 on success, negative LVL_E* otherwise (lvl_last_error() gives the text);
 *     nothing is thrown across the boundary
 *   - head dimension is fixed at 64 (all CLIP_OPENAI_TIMESFORMER_* configs: 768/12 = 1024/16 = 64)
 */
#ifndef LAVILA_HIP_H
#define LAVILA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported. */
#pragma GCC visibility push(default)

enum { LVL_F32 = 0, LVL_BF16 = 1 };
enum { LVL_OK = 0, LVL_EINVAL = -22, LVL_ENOSYS = -38, LVL_EHIP = -5 };
enum { LVL_ATTN_SPACE = 0, LVL_ATTN_TIME = 1, LVL_ATTN_CAUSAL = 2 /* lvl_attention_fast_path only */ };
enum { LVL_EPI_BIAS = 0, LVL_EPI_BIAS_QUICKGELU = 1, LVL_EPI_QUICKGELU_BWD = 2, LVL_EPI_BIAS_RESIDUAL = 3,
       LVL_EPI_BIAS_QUICKGELU_DERIV = 4, LVL_EPI_MUL_AUX_COLSUM = 5 };
/* activations of the narrator decoder's MLPs (lvl_act_inplace) */
enum { LVL_ACT_GELU_NEW = 0, LVL_ACT_SQRELU = 1 };

/* library identification / diagnostics (host pointers) */
const char* lvl_version(void);
/* Test hook for the dynamic schedules of lvl_linear_tn / lvl_linear_wgrad (`sched` != NULL): from now on the workgroups
 * with blockIdx % mod == 1 of every such launch behave as if their compute unit had been held by another kernel for
 * the whole launch -- they start, find nothing left to do and sign off -- so that the take-over paths (tile queue /
 * chunk stealing) can be exercised deterministically on an idle GPU. mod = 0 switches it off. Results must not change
 * as long as mod does not divide 8 (blockIdx % 8 selects the XCD tile queue: an XCD whose workgroups ALL never work has
 * nobody to serve its queue -- a workgroup that is merely late serves it when it starts). */
int lvl_debug_late_workgroups(int mod);
/* Compute units the two persistent GEMM kernels (lvl_linear_tn, lvl_linear_wgrad: one workgroup per CU holding the
 * whole register file) size their grids for. 0 (default) = every CU of the device; a multiple of 8 below that leaves
 * the remaining CUs to whatever runs beside the step (e.g. the channel workgroups of an RCCL collective), which
 * otherwise force a whole extra round of tiles (tools/probe_cu_contention.py). Process-wide; call before sizing
 * workspaces (lvl_workspace_floats("linear_wgrad") depends on it). Nothing in the reference corresponds to it. */
int lvl_set_compute_units(int n);
const char* lvl_last_error(void);
/* number of float32 workspace elements a call needs (host-side query, no device work) */
int64_t lvl_workspace_floats(const char* op, int64_t rows, int64_t cols);

/* ---- LayerNorm (optionally fused with the residual add that feeds it) ------------------------------
 * replaces F.layer_norm in ln_pre (timesformer.py:263-264,365-366; eps 1e-5), norm1/2/3 + norm
 * (timesformer.py:247,153,166,169,277,377; eps 1e-6), text ln_1/ln_2/ln_final
 * (openai_model.py:187,193; models.py:106,156; eps 1e-5), and -- when x2/xbias are given -- the
 * residual adds of SpaceTimeBlock.forward (timesformer.py:183,192,196) and
 * ResidualAttentionBlock.forward (openai_model.py:206-216) that produce the LayerNorm input.
 *
 * forward:  s = x (+ x2) (+ xbias);  y = (s - mean(s)) * rstd(s) * gamma + beta
 *   x, x2 (nullable), y: [rows, cols] dtype; xbias (nullable), gamma, beta: [cols] f32;
 *   s_out (nullable): [rows, cols] dtype, receives s (rounded to dtype; the statistics then use the
 *   rounded value so forward and backward agree); mean, rstd (nullable): [rows] f32.
 * backward: with s recomputed as x (+ x2) (+ xbias) exactly as in forward (pass the saved s_out as x
 *   and x2 = xbias = NULL when it was kept):
 *   dx = rstd * (dy*gamma - mean(dy*gamma) - shat * mean(dy*gamma*shat)) (+ dadd)
 *   dgamma = sum_rows dy*shat, dbeta = sum_rows dy, dxsum (nullable) = sum_rows dx (= d xbias).
 *   dadd (nullable): [rows, cols] dtype, extra gradient arriving at s through s_out.
 *   dx_plain (nullable): when given, receives the normalisation's own input gradient WITHOUT dadd (and dxsum sums
 *   that one), while dx = dx_plain + dadd: the two-consumer case of the odd residual wiring of SpaceTimeBlock
 *   (timesformer.py:183-196: x feeds x + time_out AND x + space_out), one pass instead of a separate add.
 * cols % 8 == 0, cols <= 4096. bwd workspace: lvl_workspace_floats("layernorm_bwd", rows, cols). */
int lvl_layernorm_fwd(const void* x, const void* x2, const float* xbias, const float* gamma,
                      const float* beta, void* s_out, void* y, float* mean, float* rstd,
                      int64_t rows, int cols, float eps, int dtype, void* stream);
int lvl_layernorm_bwd(const void* dy, const void* x, const void* x2, const float* xbias,
                      const float* gamma, const float* mean, const float* rstd, const void* dadd,
                      void* dx, void* dx_plain, float* dgamma, float* dbeta, float* dxsum, float* ws,
                      int64_t rows, int cols, int dtype, void* stream);

/* ---- bias + QuickGELU --------------------------------------------------------------------------
 * a = (u + bias) * sigmoid(1.702 (u + bias)); replaces the bias add of Mlp.fc1 / mlp.c_fc and
 * QuickGELU.forward (openai_model.py:177-179; timesformer.py:52-54). u,a,da,du: [rows, cols] dtype;
 * bias,dbias: [cols] f32 (bias may be NULL -> treated as 0, dbias may be NULL). cols % 8 == 0.
 * bwd workspace: lvl_workspace_floats("bias_quickgelu_bwd", rows, cols). */
int lvl_bias_quickgelu_fwd(const void* u, const float* bias, void* a, int64_t rows, int cols,
                           int dtype, void* stream);
int lvl_bias_quickgelu_bwd(const void* da, const void* u, const float* bias, void* du, float* dbias,
                           float* ws, int64_t rows, int cols, int dtype, void* stream);

/* ---- patch embedding, gather side ----------------------------------------------------------------
 * video [B,C,F,H,W] f32 (frame_major = 0: the batch contract of datasets.py:360-387, what
 * SpaceTimeTransformer.forward receives, timesformer.py:384-390) or [B,F,C,H,W] f32 (frame_major = 1:
 * what forward_features receives, timesformer.py:345-348, the narrator's entry narrator.py:74)
 * -> patch matrix [B*F*N, C*P*P] dtype, rows frame-major then (py,px), columns (c,i,j) = Conv2d weight
 * flattening. Replaces permute(0,2,1,3,4).contiguous() (timesformer.py:387) and the im2col half of
 * nn.Conv2d(k=stride=P) (timesformer.py:77,79-84); the contraction with the [D, C*P*P] weight is a
 * plain GEMM. H % P == 0, W % P == 0. */
int lvl_patchify(const float* video, void* patches, int B, int C, int F, int H, int W, int P,
                 int frame_major, int dtype, void* stream);

/* ---- token assembly: cls concat + positional/temporal embedding add --------------------------------
 * x[b,0,:] = cls + pos[0];  x[b,1+f*N+n,:] = pe[b,f*N+n,:] + pos[1+n] + temporal[f]
 * (timesformer.py:353-364). pe: [B,F*N,D] dtype; cls: [D], pos: [N+1,D], temporal: [>=F,D] f32;
 * x: [B,1+F*N,D] dtype. D % 8 == 0. */
int lvl_embed_tokens_fwd(const void* pe, const float* cls, const float* pos, const float* temporal,
                         void* x, int B, int F, int N, int D, int dtype, void* stream);
/* lvl_embed_tokens_bwd (round 5): the parameter gradients of the token assembly from dx [B, 1 + F N, D] (autograd of
 * timesformer.py:353-366): dpos [N + 1, D] f32 (row 0 = the cls position = d cls_token, row 1 + n summed over batch and
 * frames), dtem [tem_rows, D] f32 (row f summed over batch and locations; rows >= F zero). One pass over dx + a small
 * second stage; d(patch embeddings) is dx[:, 1:] itself. ws: lvl_embed_tokens_bwd_ws(F, N, D) floats. */
int64_t lvl_embed_tokens_bwd_ws(int F, int N, int D);
int lvl_embed_tokens_bwd(const void* dx, float* dpos, float* dtem, float* ws, int B, int F, int N, int D, int tem_rows,
                         int dtype, void* stream);

/* ---- text tower: token + positional embedding (round 6) ------------------------------------------------
 * x[b, l, :] = table[tokens[b, l], :] + pos[l, :]   (CLIP.encode_text, models.py:152-153: `self.token_embedding(text)`
 * + `self.positional_embedding`; under autocast the sum is rounded once to `dtype`). tokens: int64, row stride
 * tok_stride elements (a `text[:, :L]` view of the [B, 77] batch is read in place), ids clamped to [0, V); table [V, W],
 * pos [>= L, W] f32; x [B, L, W] dtype. W % 4 == 0 (forward), W % 8 == 0 (backward).
 * lvl_text_embed_bwd: autograd of the same two lines -- d table [V, W] f32 (= nn.Embedding's dense backward: row v is the
 * sum of the dx rows whose token is v, added in a fixed order -- eight contiguous row ranges, ascending inside each; rows of
 * unused ids zero) and d pos [ctx, W] f32
 * (row l = sum over the batch, rows >= L zero) -- without a sort, without float atomics and without memset nodes (torch's
 * embedding_dense_backward sorts with rocPRIM above 3072 rows, whose histogram memsets become unreliable memset NODES in a
 * replayed hipGraph). ws: lvl_text_embed_bwd_ws(B, L, V) int32 words. W <= 2048. */
int lvl_text_embed_fwd(const int64_t* tokens, int64_t tok_stride, const float* table, const float* pos, void* x, int B,
                       int L, int W, int V, int dtype, void* stream);
int64_t lvl_text_embed_bwd_ws(int B, int L, int V);
int lvl_text_embed_bwd(const void* dx, const int64_t* tokens, int64_t tok_stride, float* dtable, float* dpos, int* ws,
                       int B, int L, int W, int V, int ctx, int dtype, void* stream);

/* ---- divided space-time attention core ---------------------------------------------------------------
 * Everything in VarAttention.forward between the qkv Linear and the proj Linear
 * (timesformer.py:110-140 incl. attn() :35-39): head split, q *= 64^-0.5, CLS query over all T
 * keys, patch queries grouped per frame (LVL_ATTN_SPACE: N queries x [cls + N] keys) or per
 * location (LVL_ATTN_TIME: F queries x [cls + F] keys), softmax in f32, heads merged.
 * qkv: [B,T,3*H*64] dtype (T = 1+F*N; q|k|v thirds, head-major inside each third);
 * out/dout: [B,T,H*64] dtype; lse: [B,H,T] f32 (log-sum-exp of every query row, saved for backward);
 * dqkv: [B,T,3*H*64] dtype. Workspaces (f32): fwd lvl_workspace_floats("divided_attn_fwd", B*H, T)
 * (partial records of the CLS row), bwd lvl_workspace_floats("divided_attn_bwd", B*H, T) (delta + up to 64 partial
 * records [192] f32 per (b, h) of the cls token's d(q|k|v), one per frame / location chunk, summed in slot order: the
 * backward has no floating-point atomics and is run-to-run bit-identical since round 6). */
int lvl_divided_attn_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N,
                         int H, int mode, int dtype, void* stream);
int lvl_divided_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                         void* dqkv, float* ws, int B, int F, int N, int H, int mode, int dtype,
                         void* stream);

/* host-side query: 1 if a bf16 call of this shape runs on the MFMA / register-tiled attention kernels (forward and
 * backward), 0 if it falls to the shape-generic kernels (any shape, correct, much slower). mode: LVL_ATTN_SPACE /
 * LVL_ATTN_TIME with (F, N, H), or LVL_ATTN_CAUSAL with N = L. The Python layer logs one warning per slow shape. */
int lvl_attention_fast_path(int mode, int F, int N, int H);
/* The same query for FLOAT32 tensors (the parity configuration, north_star "within 1e-3 fp32"): 1 if the call runs on
 * the f32-class instantiations of the fast kernels -- the MFMA kernels with every operand as hi/lo bf16 images and three
 * MFMAs per product (space groups up to 272 / 288 keys forward / backward, causal text up to 272 / 256 tokens), the
 * register-tiled time kernels with float32 rows (1-4, 8, 16 frames) -- 0 if it falls to the shape-generic kernels.
 * lvl_debug_f32_generic(1) (or LAVILA_F32_GENERIC=1 in the environment) sends every float32 attention call to the
 * generic kernels: the A/B switch of the parity tests. */
int lvl_attention_fast_path_f32(int mode, int F, int N, int H);
int lvl_debug_f32_generic(int on);
/* Space groups of more than 288 keys (TSF-L/14 at 336: 577 per frame; float32: more than 272) run on KEY-TILED STREAMING
 * kernels (csrc/attn_space_stream.hip: a workgroup owns 128 queries -- or keys, in the dK/dV kernel -- and streams the
 * other side through double-buffered 64-row LDS images; online softmax forward, lse-recomputed P backward) instead of
 * keeping the whole group LDS-resident with one workgroup per compute unit. Test / measurement hook: mode 1 = streaming
 * kernels for EVERY space group, -1 = never (the LDS-resident kernels up to 592 keys, as in round 3), 0 = the shipped
 * choice. Results agree to rounding (bf16) / f32 summation order. */
int lvl_debug_space_stream(int mode);
/* Measurement hook, bf16 streaming kernels: bit 0 = the forward's 3-workgroup cut with a three-stage LDS-DMA ring (default:
 * 4 workgroups per compute unit, two stages); bit 2 = register staging instead of the LDS-DMA rings (all three kernels;
 * bit 0 then selects the forward's 4-workgroup register-staged cut). Same results. */
int lvl_debug_stream_variant(int v);
/* The "fp8 MFMA QK^T path" BASELINE.json names for TSF-L/14 at 336 (configs[3]): on = the streaming space kernels (bf16
 * tensors) compute their score products q.k with v_mfma_f32_16x16x32_fp8_fp8 on q / k fragments rounded to OCP e4m3 in
 * registers (saturating at +-448), forward AND the backward's recomputation of P, so lse and P stay consistent; every
 * other product (P V, dP, dQ, dK, dV) stays a bf16 MFMA, tensors stay bf16. Off by default (LAVILA_FP8_QK=1 in the
 * environment turns it on at first use): e4m3 scores cost accuracy and, at head dim 64, buy no time.
 * Groups the streaming kernels do not take (<= 288 keys, unless lvl_debug_space_stream(1)) and float32 tensors are
 * unaffected. */
int lvl_set_fp8_qk(int on);
/* Test hook: how many lvl_divided_attn_* / lvl_causal_attn_* calls of this process were served by the shape-generic
 * kernels so far (reset != 0: read and clear). */
int lvl_debug_generic_attention_calls(int reset);

/* ---- causal self-attention core of the text tower -------------------------------------------------------
 * nn.MultiheadAttention core with the additive causal mask (openai_model.py:196-198,
 * models.py:131-137) on packed qkv [B,L,3*H*64] (batch-major; the reference's LND permutes
 * models.py:153,155 are folded away). out: [B,L,H*64]; lse: [B,H,L].
 * bwd workspace: lvl_workspace_floats("causal_attn_bwd", B*H, L). */
int lvl_causal_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dtype,
                        void* stream);
int lvl_causal_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                        void* dqkv, float* ws, int B, int L, int H, int dtype, void* stream);

/* ---- contrastive (InfoNCE) head ------------------------------------------------------------------------------
 * Slab formulation of CLIPLoss.forward (loss.py:69-118): this rank owns global rows
 * [row0, row0+B). Direction 0: logits_per_image rows = (scale*img_local) @ txt_all^T; direction 1:
 * logits_per_text rows = (scale*txt_local) @ img_all^T (loss.py:78-79,92-93).
 * img_all/txt_all: [G,E] dtype (rank-ordered all-gather, distributed_utils.py:88); the local slabs
 * are rows [row0,row0+B) of them. E % 8 == 0. scale: DEVICE pointer to exp(logit_scale) (1 float, so
 * the host never synchronises on it).
 * stats: [2,B,4] f32 = {lse, diag logit, sum_j softmax_j * logit_j, max logit};
 * argmax: [2,B] int32 (first maximal column -- torch.argmax tie rule; loss.py:113);
 * logits (optional, may be NULL): [2,B,G] f32 slab of the scaled logits (for parity checks).
 * bwd: lse_all [2,G] f32 = gathered row LSEs of both directions; upstream (nullable): DEVICE pointer
 * to d(objective)/d(loss); coef: host factor mult/(2G). Writes coef*upstream*d(sum of both CE
 * sums)/d(img_local | txt_local): [B,E] f32. No gradient collective is needed.
 * rows_only != 0: CLIPLoss(local_loss=True) without gather_with_grad (loss.py:34-43,86-88): the gathered
 * partner rows are constants, so each local row receives only the gradient of its own two cross-entropies
 * (only the local entries of lse_all are read; coef = 1/(2B)). */
int lvl_clip_loss_fwd(const void* img_all, const void* txt_all, const float* scale, int B, int G,
                      int E, int row0, float* stats, int32_t* argmax, float* logits, int dtype,
                      void* stream);
int lvl_clip_loss_bwd(const void* img_all, const void* txt_all, const float* lse_all,
                      const float* scale, const float* upstream, float coef, int B, int G, int E,
                      int row0, int rows_only, float* dimg, float* dtxt, int dtype, void* stream);

/* ---- contrastive head with per-pair temperature (SSLCLIPLoss, loss.py:121-217) ---------------------------
 * Same slab structure as lvl_clip_loss_*; ind_all: [G] int32 gt_indicators in rank order (1 = ground-truth
 * narration, 0 = pseudo-label); scales3: DEVICE pointer to {pseudo, sqrt(pseudo*real), real}; the pair (i,j)
 * uses scales3[ind[i] + ind[j]] (loss.py:160-166,172-178).
 * stats: [2,B,8] f32 = {lse, diag logit, E0, E1, E2, diag dot, max logit, 0} with
 * Ek = sum_{j: ind[i]+ind[j]==k} softmax_j * (a_i . b_j): d(loss)/d(scales3[k]) follows without another pass. */
int lvl_ssl_clip_loss_fwd(const void* img_all, const void* txt_all, const int32_t* ind_all,
                          const float* scales3, int B, int G, int E, int row0, float* stats,
                          int32_t* argmax, float* logits, int dtype, void* stream);
int lvl_ssl_clip_loss_bwd(const void* img_all, const void* txt_all, const int32_t* ind_all,
                          const float* lse_all, const float* scales3, const float* upstream, float coef,
                          int B, int G, int E, int row0, float* dimg, float* dtxt, int dtype,
                          void* stream);

/* ---- cls-only attention (last block of a cls-pooled forward) ---------------------------------------------------
 * When only `norm(x)[:, 0]` leaves the tower (SpaceTimeTransformer.forward, timesformer.py:377,384-390) the last
 * block's space attention is needed for its cls query alone, which attends to all T tokens (timesformer.py:116-119):
 *   out[b,h,:] = softmax_j(0.125 * q[b,h,:] . k[b,j,h,:]) v[b,j,h,:]
 * q: [B, H*64] (the q third of the qkv Linear applied to the cls rows), kv: [B, T, 2*H*64] (its k and v thirds applied
 * to all rows), out: [B, H*64], lse: [B, H] f32 (natural log of the softmax denominator of the scaled scores).
 * Backward: dq [B, H*64] f32, dkv [B, T, 2*H*64] dtype, from dout [B, H*64]. */
int lvl_cls_attn_fwd(const void* q, const void* kv, void* out, float* lse, int B, int T, int H, int dtype, void* stream);
int lvl_cls_attn_bwd(const void* q, const void* kv, const void* out, const void* dout, const float* lse, float* dq,
                     void* dkv, int B, int T, int H, int dtype, void* stream);

/* ---- narrator seam: multi-query cross-attention pooling (inference) ------------------------------------------
 * The core of coca.py's CrossAttention (lavila/models/coca.py:93-123) between to_q / to_kv and to_out, as
 * VCLM_HF.encode_image runs it on the tower's token features (narrator.py:44-49,88-90): NQ x H query rows of 64
 * channels attend to the Tk context tokens through ONE shared key/value head:
 *   out[b,n,h,:] = softmax_j(0.125 * q[b,n,h,:] . k[b,j,:]) v[b,j,:]   (q * dim_head^-0.5, max-subtracted softmax)
 * q: [B or 1, NQ, H*64] with batch stride q_batch_stride elements (0 = the same queries for every clip: the narrator
 * repeats its learned img_queries); kv: [B, Tk, 128] = k | v as to_kv writes them; out: [B, NQ, H*64]. Forward only. */
int lvl_mq_cross_attn_fwd(const void* q, int64_t q_batch_stride, const void* kv, void* out, int B, int NQ, int H,
                          int Tk, int dtype, void* stream);

/* ---- narrator decoder: gated-cross-attention GPT-2 (inference) --------------------------------------------------
 * The row passes of lavila/models/gpt2_gated.py's GPT2LMHeadModel around its Conv1D GEMMs (which run on lvl_linear_tn
 * against [out,in] bf16 copies of the [in,out] Conv1D weights). The reference's VCLM_HF.generate re-runs the whole prefix
 * for every token (narrator.py:118-143, use_cache=False); this ABI decodes ONE row per sequence and step against a
 * key/value cache whose fill level `pos_dev` is an int32 in DEVICE memory, so one captured hipGraph serves every step.
 * dtype = element type of activations, caches and embedding tables (LVL_F32 / LVL_BF16); arithmetic is f32.
 *
 * lvl_gpt2_embed (gpt2_gated.py:892-895): out[r,:] = wte[ids[r],:] + wpe[p0 + r % L,:], p0 = *pos_dev (0 when NULL);
 *   ids [rows] int64 (clamped into the table), wte [vocab, D], wpe [positions, D], D % 8 == 0.
 * lvl_gated_add_layernorm: s = res + (*gate) * y (gate NULL = 1, y NULL = no add), h = LayerNorm(s) * gamma + beta with
 *   biased variance over the D channels -- the residual adds of GPT2Block.forward (gpt2_gated.py:442-458 with the
 *   tanh(alpha) gates, :475, :483-487) fused with the LayerNorm that reads the sum next (ln_2_crossattention, ln_1, ln_2,
 *   the next block's first norm, ln_f). res / y / s / h: [rows, D]; s may be res (in place) or NULL; gate: one f32
 *   (tanh(alpha), computed by the caller); gamma / beta [D] f32; D % 8 == 0, D <= 4096.
 * lvl_act_inplace (gpt2_gated.py:363-396): u <- gelu_new(u) = 0.5 u (1 + tanh(sqrt(2/pi) (u + 0.044715 u^3))) or
 *   relu(u)^2 (mlp_crossattention), n % 8 == 0.
 * lvl_decode_self_attn (gpt2_gated.py:206-238,337-345 for one new token): qkv [B, 3*H*64] = q | k | v of the new token;
 *   cache [B, Tcap, 2*H*64] = k | v rows of the tokens so far. p = *pos_dev: row p of the cache is WRITTEN with this
 *   step's k | v, then out[b,h,:] = softmax_{j<=p}(0.125 q . k_j) v_j (the causal mask of a last-row query = all rows
 *   so far). The caller advances *pos_dev after the last layer.
 * lvl_cross_attn_rows_fwd (gpt2_gated.py:327-334 + _attn without a mask): multi-head attention of `rows` independent
 *   query rows q [rows, H*64] over per-context keys / values kv [rows/qrep, Tk, 2*H*64] = k | v (the cross-attention
 *   c_attn applied to the image tokens once per clip); qrep consecutive query rows share one context (the L positions
 *   of a teacher-forced caption, or the num_return_sequences samples of a clip). out [rows, H*64]. */
/* lvl_linear_skinny: y[M,N] = act(x[M,K] . w[N,K]^T + bias) for FEW rows -- the decoder's Conv1Ds while decoding
 * (M = captions in flight; gpt2_gated.py:327-334,337,354,392-394) and its lm_head. bf16 x / w / y, f32 bias (nullable)
 * and accumulation; act: -1 none, LVL_ACT_GELU_NEW, LVL_ACT_SQRELU (applied to the f32 sum before the bf16 store).
 * Up to 128 rows: one workgroup per 16 rows x 16 (N < 2048) or 32 x 64 columns, its 8 waves splitting K, fragments
 * straight from memory (lvl_linear_tn gives a 256-column panel to one compute unit, which at M <= 64 leaves the chip idle).
 * Beyond 128 rows, and for N >= 8192 (lm_head), K % 64 == 0: LDS-staged 64 x 64 / 64 x 128 tiles (whole-line loads, three K
 * blocks in flight). N % 16 == 0 and K % 32 == 0, else LVL_ENOSYS; any M >= 0. */
int lvl_linear_skinny(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, int act, void* stream);
/* lvl_linear_skinny_f32c (round 5): the same product for FLOAT32 operands in f32-class mode -- x3 [M, K3] and w3 [N, K3]
 * are the bf16 term images of the float32 x [M, K] / w [N, K] (K3 = 3 K; lvl_split_bf16x3 role 0 for x, role 1 for w:
 * h|h|l against h|l|h, so one pass over K3 accumulates h.h' + h.l' + l.h'), y [M, N] float32, bias float32, activation on
 * the unrounded sum. Replaces the float32 `x @ W + b` of the reference's Conv1D / nn.Linear (gpt2_gated.py:184-188,
 * 383-384,1010; coca.py:78-88) where the widths are not multiples of 256 (lvl_linear_tn's f32-class mode takes those):
 * the narrator's float32 decoder, its lm_head and the float32 inference Linears of small towers. ~2^-17 relative per
 * product. N % 16 == 0 and K % 32 == 0, else LVL_ENOSYS. */
int lvl_linear_skinny_f32c(const void* x3, const void* w3, const float* bias, float* y, int M, int N, int K3, int act,
                           void* stream);
/* lvl_linear_skinny_ln: out[M,N] = act(LayerNorm(res + (*gate) * y) . w^T + bias), res_out = bf16(res + (*gate) * y) --
 * lvl_gated_add_layernorm folded into the prologue of the Conv1D that consumes it (q_attn / c_attn / the two c_fc of a
 * GPT2Block, gpt2_gated.py:441-487): a workgroup of lvl_linear_skinny owns whole rows of its input, so it forms the sum,
 * the row statistics and the normalised operand in registers before its first MFMA. y NULL = no add (res_out unused);
 * gate NULL = 1; res_out must NOT be res (other column strips are still reading it). gamma / beta [K] f32.
 * N % 16 == 0, K % 32 == 0, K <= 1792, else LVL_ENOSYS; meant for M <= 128 (decoding). */
int lvl_linear_skinny_ln(const void* res, const void* y, const float* gate, const float* gamma, const float* beta,
                         float eps, void* res_out, const void* w, const float* bias, void* out, int M, int N, int K,
                         int act, void* stream);
/* Measurement hook (tools/probe_skinny.py): selects another workgroup tiling / k-step assignment of lvl_linear_skinny for
 * shapes that allow it (0 = the shipped choice). Results are the same up to f32 summation order. */
int lvl_debug_skinny_variant(int variant);
int lvl_gpt2_embed(const int64_t* ids, const void* wte, const void* wpe, const int* pos_dev, void* out, int rows, int L,
                   int D, int vocab, int positions, int dtype, void* stream);
int lvl_gated_add_layernorm(const void* res, const void* y, const float* gate, const float* gamma, const float* beta,
                            float eps, void* sum_out, void* h_out, int rows, int D, int dtype, void* stream);
int lvl_act_inplace(void* u, int64_t n, int act, int dtype, void* stream);
int lvl_decode_self_attn(const void* qkv, void* cache, const int* pos_dev, void* out, int B, int Tcap, int H, int dtype,
                         void* stream);
int lvl_cross_attn_rows_fwd(const void* q, const void* kv, void* out, int rows, int qrep, int Tk, int H, int dtype,
                            void* stream);
/* Measurement hook: n = 4 / 8 / 16 forces lvl_cross_attn_rows_fwd's VALU shared-context kernel with n waves per
 * workgroup (-1: 16) where the MFMA kernel would run (bf16, qrep >= 2, Tk <= 256); 0 = the shipped choice. */
int lvl_debug_cross_attn_waves(int waves);

/* lvl_sample_next_token: everything VCLM_HF.generate does with one step's logits (narrator.py:122-137 and the warpers
 * of :368-389 = transformers' Temperature / TopK / TopP logits warpers with min_tokens_to_keep = 1), one workgroup per
 * caption, the row resident in LDS as 16-bit keys -- no sort, no [rows, vocab] temporaries:
 *   nll[r]     = target ? (target[r] == pad_id ? 0 : logsumexp(l_r) - l_r[target[r]])      F.cross_entropy(ignore_index=pad)
 *                       : entropy of softmax(l_r)                                          torch.special.entr(softmax).sum()
 *   counted[r] = target ? (target[r] != pad_id) : 1
 *   next_token[r] ~ softmax(warp(l_r)): l / temperature; keep the top_k largest (0 = off; ties with the k-th stay);
 *                drop the ascending-probability tail whose cumulative mass is <= 1 - top_p (1 = off), always keeping
 *                the largest; inverse CDF over the kept entries in index order at uniform[r] (in [0,1), e.g. torch.rand).
 *                top_k = 1 is greedy: the first maximum, whatever the uniform.
 * logits: [rows, >= vocab] bf16, row stride row_stride elements (a multiple of 8, rows 16-byte aligned, >= vocab rounded
 * up to 8: the padded product the lm_head GEMM leaves). vocab <= lvl_sample_max_vocab() (the row must fit 160 KB of
 * LDS; GPT-2's 50257 does), else LVL_ENOSYS. target / dbg nullable; dbg [rows,12] f32 = (lowest kept value, ties dropped
 * at the top-p boundary, kept mass relative to e^max, top-p boundary value; microseconds spent in the kernel's 5 phases,
 * start time) for tests and tools/probe_narrator.py. */
int lvl_sample_max_vocab(void);
int lvl_sample_next_token(const void* logits, int64_t row_stride, int rows, int vocab, float temperature, int top_k,
                          float top_p, const float* uniform, const int64_t* target, int64_t pad_id, int64_t* next_token,
                          float* nll, float* counted, float* dbg, void* stream);

/* ---- Linear layers: forward and input-gradient GEMMs with fused epilogues -------------------------------------
 * y[M,N] = epilogue(x[M,K] . w[N,K]^T): both operands bf16, row-major, contraction-contiguous; f32 accumulation.
 * Forward of every nn.Linear on the path (qkv/proj timesformer.py:95-96,110,142; Mlp fc1/fc2 timesformer.py:47-58;
 * the Conv2d(k=s=P) contraction timesformer.py:77,83; text in_proj/out_proj/c_fc/c_proj openai_model.py:186-192)
 * with w = the bf16 weight [out,in]; input gradient dx = dy . wt^T with wt = the transposed bf16 copy [in,out]
 * that lvl_cast_transpose writes (what autograd computes for the Linear input when loss.backward() runs,
 * main_pretrain.py:520). bias: [N] f32, nullable. Epilogues:
 *   LVL_EPI_BIAS             y = acc + bias
 *   LVL_EPI_BIAS_QUICKGELU   aux_out = u = bf16(acc + bias); y = u * sigmoid(1.702 u)     (fc1 + QuickGELU,
 *                            timesformer.py:52-54, openai_model.py:177-179; u is kept for the backward)
 *   LVL_EPI_QUICKGELU_BWD    y = acc * d quickgelu(aux_in); colsum[N] f32 = column sums of y (= d fc1.bias);
 *                            acc = dA = dY . W2 is the input gradient of fc2, y = d(fc1 output)
 *   LVL_EPI_BIAS_RESIDUAL    y = acc + bias + aux_in: the Linear's output added to the residual stream, in f32 before the
 *                            one rounding of the sum -- `x + attn(norm1(x))`, `x + mlp(norm2(x))` (timesformer.py:183-196;
 *                            openai_model.py:199-200) leave proj / fc2 as the new stream; aux_in [M,N], y's dtype
 *   LVL_EPI_BIAS_QUICKGELU_DERIV   (bf16 only) u = acc + bias (f32, not rounded); y = u * sigmoid(1.702 u);
 *                            aux_out = d quickgelu(u) = s (1 + 1.702 u (1 - s)), s = sigmoid(1.702 u): the training form of
 *                            LVL_EPI_BIAS_QUICKGELU -- autograd's backward of timesformer.py:52-54 needs the derivative,
 *                            not u, and the forward holds s in a register (one exp2 + one reciprocal serve both outputs)
 *   LVL_EPI_MUL_AUX_COLSUM   (bf16 only) y = acc * aux_in; colsum[N] f32 = column sums of y: LVL_EPI_QUICKGELU_BWD on the
 *                            stored derivative (no transcendental in the backward's epilogue)
 * aux_out / aux_in: [M,N] bf16. N % 256 == 0 and K % 64 == 0 (operands < 4 GiB), else LVL_ENOSYS. Workspace (QUICKGELU_BWD /
 * MUL_AUX_COLSUM only): lvl_workspace_floats("linear_tn", M, N) floats.
 * dtype = LVL_F32 selects the F32-CLASS MODE of the same kernel (the parity configuration, north_star "within 1e-3
 * fp32"): x [M, 3K0] and w [N, 3K0] are the bf16 term images lvl_split_bf16x3 writes (role 0 for x, role 1 for w;
 * K = 3*K0, K0 % 64 == 0), so that one pass accumulates xh.wh + xh.wl + xl.wh in f32 (~2^-17 relative per product);
 * y, aux_out and aux_in are FLOAT32 [M,N] and the QuickGELU epilogues see the unrounded pre-activation. Same tiling,
 * LDS-DMA ring, MFMA schedule, bias path and tile schedule as the bf16 mode.
 * sched (nullable): the launch's TILE-COUNTER block, 16 x uint32 (64-byte aligned), ZERO on entry; the kernel leaves it
 * zero again when its last workgroup exits, so a caller may hand the same block to the next launch on the SAME stream
 * (launches that can run concurrently -- other streams, graph branches -- need distinct blocks). With it the persistent
 * workgroups take their 256x256 tiles from per-XCD device counters instead of static ranges: a compute unit held by
 * another kernel (an RCCL channel during DDP's gradient all-reduce, main_pretrain.py:180-183) then costs one tile, not
 * a range. NULL = static ranges (identical results: the schedule does not change any tile's arithmetic). */
int lvl_linear_tn(const void* x, const void* w, const float* bias, void* y, void* aux_out, const void* aux_in,
                  float* colsum, float* ws, uint32_t* sched, int64_t M, int N, int K, int epilogue, int dtype,
                  void* stream);

/* ---- Linear-layer weight gradient ------------------------------------------------------------------------
 * dW[N,K] = dY[M,N]^T X[M,K], dbias[N] (nullable) = column sums of dY: what torch.autograd computes for the weight
 * and bias of every nn.Linear on the path (qkv/proj: timesformer.py:96-99, Mlp fc1/fc2: timesformer.py:47-50) when
 * `loss.backward()` runs (main_pretrain.py:520). dy: [M,N], x: [M,K] bf16 row-major; dw: [N,K] f32; dbias: [N] f32.
 * Tiled for N, K multiples of 192/288/384 (TSF-B/L widths); other shapes return LVL_ENOSYS and the caller keeps
 * the library GEMM. Workspace: lvl_workspace_floats("linear_wgrad", N, K) floats (-1 = unsupported shape).
 * sched (nullable): CHUNK-COUNTER block of the launch, 1024 x uint32 (64-byte aligned), ZERO on entry, zero again when
 * the launch has drained (same reuse rule as lvl_linear_tn's block). With it every (tile, row split) unit is cut into
 * row chunks handed out by device counters: a workgroup works through its own unit and then takes unclaimed chunks of
 * the other splits of its tile, so a compute unit held by another kernel delays the launch by a fraction of a unit
 * instead of a second round. With all CUs available nobody steals and the result is bit-identical to the static plan
 * (NULL). f32-class weight gradients use the same entry with row-STACKED term images (lvl_split_bf16x3 with
 * dst_term_stride = rows_padded * cols): dy3 [3M,N] = (h; h; l), x3 [3M,K] = (h; l; h); dw is float32 either way. */
int lvl_linear_wgrad(const void* dy, const void* x, float* dw, float* dbias, float* ws, uint32_t* sched, int64_t M,
                     int N, int K, int dtype, void* stream);

/* ---- bias gradient of the qkv Linear that feeds an attention core ------------------------------------------------
 * dbias[3D] = column sums over all rows of the dqkv an attention backward produced (what autograd computes for
 * `qkv.bias`, timesformer.py:96 / `in_proj_bias`, openai_model.py:196-198) without reading the k and v thirds:
 * every softmax row sums to 1, so sum_rows(dv) = sum_rows(dout); the scores are invariant to a constant added to
 * every key, so sum_rows(dk) = 0; only the q third of dqkv is reduced. dqkv: [rows, 3D], dout: [rows, D] dtype;
 * dbias: [3D] f32. Workspace: lvl_workspace_floats("qkv_bias_grad", rows, D). */
int lvl_qkv_bias_grad(const void* dqkv, const void* dout, float* dbias, float* ws, int64_t rows, int D, int dtype,
                      void* stream);
/* lvl_divided_attn_bwd_bias (round 5): lvl_divided_attn_bwd AND the qkv-bias gradient of the same Linear in one call,
 * without the pass over dqkv / dout where the kernels can avoid it (autograd of timesformer.py:96,110-140 wrt qkv.bias):
 *   q third -- the LDS-resident fused space kernel and the register-tiled time kernels hold the dQ accumulators / dq
 *              rows in registers: they write per-workgroup column sums (a [B F, D] / [B chunks, D] f32 slab) on the way
 *              and a two-stage column reduce finishes them; other kernel families reduce the q third of dqkv afterwards;
 *   k third -- exactly 0;
 *   v third -- `dout_colsum` [D] f32 when the caller has it (dout is the input gradient dy.W of the projection Linear:
 *              sum_rows(dout) = sum_rows(dy).W = d(b_proj).W, lvl_vec_mat_f32), else (NULL) reduced here from dout.
 * dbias [3D] f32; ws as for lvl_divided_attn_bwd; ws2: lvl_divided_attn_bwd_bias_ws(...) floats. */
int64_t lvl_divided_attn_bwd_bias_ws(int B, int F, int N, int H, int mode, int dtype);
int lvl_divided_attn_bwd_bias(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                              const float* dout_colsum, float* dbias, float* ws2, int B, int F, int N, int H, int mode,
                              int dtype, void* stream);
/* Measurement / selection hook: bias-gradient rider of the register-tiled time backward kernels -- 0 none (the q third is
 * reduced from dqkv afterwards), 1 column sums in registers at the kernels' usual occupancy, 2 at one wave per SIMD less
 * (no spills), -1 (default) the measured choice per shape. Results are identical up to f32 summation order. */
int lvl_debug_time_bwd_rider(int mode);
/* lvl_vec_mat_f32: out[K] = v[N] . W[N, K] (float32, row-major W) -- column sums of a Linear's input gradient from the
 * column sums of its output gradient: dx = dy W  =>  sum_rows(dx) = sum_rows(dy) W (what feeds `dout_colsum` above). */
int lvl_vec_mat_f32(const float* v, const float* W, float* out, int N, int K, void* stream);

/* ---- weight staging of a Linear layer under bf16 autocast ----------------------------------------------------
 * dst[n,k] = bf16(src[n,k]), dst_t[k,n] = bf16(src[n,k]): the cast autocast applies to nn.Linear weights
 * (main_pretrain.py:491 `amp.autocast`) plus the transposed copy the input-gradient GEMM wants, in one pass.
 * src: [N,K] f32; dst: [N,K] bf16; dst_t: [K,N] bf16. */
int lvl_cast_transpose(const float* src, void* dst, void* dst_t, int N, int K, void* stream);
/* The same for MANY weights in one launch (every Linear weight of both towers is re-cast once per optimizer step:
 * main_pretrain.py:520-533 `optimizer.step()` then the next `model(...)`): desc = DEVICE array of `count` records of 40 bytes
 * { const float* src; void* dst; void* dst_t; uint32_t N; uint32_t K; int64_t tile0; }, tile0 = number of 64 x 64 tiles of
 * the records in front (tile0 of record 0 = 0, ascending), total_tiles = their sum over all records. */
int lvl_cast_transpose_multi(const void* desc, int count, int64_t total_tiles, void* stream);

/* ---- f32-class operands for the MFMA GEMMs (parity configuration) ----------------------------------------------
 * Writes a float32 matrix src [rows, cols] (row stride src_row_stride elements) as three bf16 TERM images:
 * h = bf16(x), l = bf16(x - h) (x = h + l up to 2^-18 |x|); image t of element (r, c) goes to
 * dst[r * dst_row_stride + t * dst_term_stride + c], t = 0..2, with the terms (h, h, l) for role 0 (the x / dy side of a
 * product) and (h, l, h) for role 1 (the w / x side): contracting image against image gives h.h' + h.l' + l.h'.
 * Replaces nothing in the reference: it is what lets `F.linear` on float32 tensors (the reference's CPU / fp32 path,
 * timesformer.py:47-58,95-99) run on lvl_linear_tn / lvl_linear_wgrad instead of a library GEMM. cols % 4 == 0. */
int lvl_split_bf16x3(const float* src, void* dst, int64_t rows, int cols, int64_t src_row_stride,
                     int64_t dst_row_stride, int64_t dst_term_stride, int role, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* LAVILA_HIP_H */
