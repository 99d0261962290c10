// C-ABI dispatch for the attention entry points: picks the MFMA / register-tiled fast kernels when
// the shape and dtype allow, otherwise the shape-generic f32-arithmetic kernels (attn_generic.hip).
#include "common.h"

int lvl_generic_divided_fwd(const void* qkv, void* out, float* lse, int B, int F, int N, int H, int mode, int dtype,
                            hipStream_t st, bool do_groups, bool do_cls);
int lvl_generic_divided_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                            float* ws, int B, int F, int N, int H, int mode, int dtype, hipStream_t st);
int lvl_generic_causal_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dtype, hipStream_t st);
int lvl_generic_causal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                           float* ws, int B, int L, int H, int dtype, hipStream_t st);
bool lvl_space_mfma_supported(int F, int N);
int lvl_space_mfma_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st);
bool lvl_time_fast_supported(int F, int N, int H);
bool lvl_time_mfma_supported(int F, int N, int H);
int lvl_time_mfma_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st);
int lvl_time_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int F, int N, int H, hipStream_t st);
// 5..16 frames: the MFMA kernels (attn_time_mfma.hip; 4.2 TB/s forward at 16 frames against 1.0 TB/s for the
// register-tiled kernels, profiles/r02_time_attention_mfma_vs_valu.txt); up to 4 frames, or a head count that is not
// a multiple of 4: the register-tiled kernels
static bool time_use_mfma(int F, int N, int H) { return lvl_time_mfma_supported(F, N, H); }
bool lvl_text_mfma_supported(int L);
int lvl_text_mfma_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, hipStream_t st);
bool lvl_text_mfma_bwd_supported(int L);
int lvl_text_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int L, int H, hipStream_t st);
bool lvl_space_mfma_bwd_supported(int F, int N);
int lvl_space_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                       int B, int F, int N, int H, hipStream_t st);
bool lvl_time_fast_bwd_supported(int F, int N, int H);
int lvl_time_fast_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int F, int N, int H, hipStream_t st);
int lvl_time_fast_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st);

static int check_divided(const char* name, const void* qkv, const void* out, int B, int F, int N, int H, int mode,
                         int dtype) {
  LVL_REQUIRE(qkv && out, "%s: null pointer", name);
  LVL_REQUIRE(B >= 0 && F > 0 && N > 0 && H > 0, "%s: bad shape B=%d F=%d N=%d H=%d", name, B, F, N, H);
  LVL_REQUIRE(mode == LVL_ATTN_SPACE || mode == LVL_ATTN_TIME, "%s: unknown mode %d", name, mode);
  LVL_REQUIRE(dtype == LVL_F32 || dtype == LVL_BF16, "%s: unknown dtype %d", name, dtype);
  LVL_REQUIRE(lvl_aligned16(qkv) && lvl_aligned16(out), "%s: pointers must be 16-byte aligned", name);
  return LVL_OK;
}

extern "C" int lvl_divided_attn_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H,
                                    int mode, int dtype, void* stream) {
  if (int rc = check_divided("divided_attn_fwd", qkv, out, B, F, N, H, mode, dtype)) return rc;
  LVL_REQUIRE(lse && ws, "divided_attn_fwd: null lse / workspace");
  if (B == 0) return LVL_OK;
  if (dtype == LVL_BF16 && mode == LVL_ATTN_SPACE && lvl_space_mfma_supported(F, N))
    return lvl_space_mfma_fwd(qkv, out, lse, ws, B, F, N, H, (hipStream_t)stream);
  if (dtype == LVL_BF16 && mode == LVL_ATTN_TIME && time_use_mfma(F, N, H))
    return lvl_time_mfma_fwd(qkv, out, lse, ws, B, F, N, H, (hipStream_t)stream);
  if (dtype == LVL_BF16 && mode == LVL_ATTN_TIME && lvl_time_fast_supported(F, N, H))
    return lvl_time_fast_fwd(qkv, out, lse, ws, B, F, N, H, (hipStream_t)stream);
  return lvl_generic_divided_fwd(qkv, out, lse, B, F, N, H, mode, dtype, (hipStream_t)stream, true, true);
}

extern "C" int lvl_divided_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                    float* ws, int B, int F, int N, int H, int mode, int dtype, void* stream) {
  if (int rc = check_divided("divided_attn_bwd", qkv, out, B, F, N, H, mode, dtype)) return rc;
  LVL_REQUIRE(dout && lse && dqkv && ws, "divided_attn_bwd: null pointer");
  LVL_REQUIRE(lvl_aligned16(dout) && lvl_aligned16(dqkv), "divided_attn_bwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  if (dtype == LVL_BF16 && mode == LVL_ATTN_SPACE && lvl_space_mfma_bwd_supported(F, N))
    return lvl_space_mfma_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, (hipStream_t)stream);
  if (dtype == LVL_BF16 && mode == LVL_ATTN_TIME && time_use_mfma(F, N, H))
    return lvl_time_mfma_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, (hipStream_t)stream);
  if (dtype == LVL_BF16 && mode == LVL_ATTN_TIME && lvl_time_fast_bwd_supported(F, N, H))
    return lvl_time_fast_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, (hipStream_t)stream);
  return lvl_generic_divided_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, mode, dtype, (hipStream_t)stream);
}

extern "C" int lvl_causal_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dtype,
                                   void* stream) {
  LVL_REQUIRE(qkv && out && lse, "causal_attn_fwd: null pointer");
  LVL_REQUIRE(B >= 0 && L > 0 && H > 0, "causal_attn_fwd: bad shape B=%d L=%d H=%d", B, L, H);
  LVL_REQUIRE(lvl_aligned16(qkv) && lvl_aligned16(out), "causal_attn_fwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  if (dtype == LVL_BF16 && lvl_text_mfma_supported(L)) return lvl_text_mfma_fwd(qkv, out, lse, B, L, H, (hipStream_t)stream);
  return lvl_generic_causal_fwd(qkv, out, lse, B, L, H, dtype, (hipStream_t)stream);
}

extern "C" int lvl_causal_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                   float* ws, int B, int L, int H, int dtype, void* stream) {
  LVL_REQUIRE(qkv && out && dout && lse && dqkv && ws, "causal_attn_bwd: null pointer");
  LVL_REQUIRE(B >= 0 && L > 0 && H > 0, "causal_attn_bwd: bad shape B=%d L=%d H=%d", B, L, H);
  LVL_REQUIRE(lvl_aligned16(qkv) && lvl_aligned16(out) && lvl_aligned16(dout) && lvl_aligned16(dqkv),
              "causal_attn_bwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  if (dtype == LVL_BF16 && lvl_text_mfma_bwd_supported(L))
    return lvl_text_mfma_bwd(qkv, out, dout, lse, dqkv, ws, B, L, H, (hipStream_t)stream);
  return lvl_generic_causal_bwd(qkv, out, dout, lse, dqkv, ws, B, L, H, dtype, (hipStream_t)stream);
}

// 1 if a bf16 call of this shape runs on the MFMA / register-tiled kernels (forward AND backward), 0 if it lands on
// the shape-generic kernels of attn_generic.hip (correct, latency-bound). Host-side query, no device work.
extern "C" int lvl_attention_fast_path(int mode, int F, int N, int H) {
  if (mode == LVL_ATTN_SPACE) return lvl_space_mfma_supported(F, N) && lvl_space_mfma_bwd_supported(F, N);
  if (mode == LVL_ATTN_TIME)
    return lvl_time_mfma_supported(F, N, H) || (lvl_time_fast_supported(F, N, H) && lvl_time_fast_bwd_supported(F, N, H));
  if (mode == LVL_ATTN_CAUSAL) return lvl_text_mfma_supported(N) && lvl_text_mfma_bwd_supported(N);
  return 0;
}
