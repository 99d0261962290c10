// C-ABI dispatch for the attention entry points: picks the MFMA / register-tiled fast kernels when
// the shape allows, otherwise the shape-generic f32-arithmetic kernels (attn_generic.hip). float32 tensors (the parity
// configuration) run on the SAME fast kernels as bf16: the MFMA kernels in their split-operand f32-class instantiation
// (PrecSplit, attn_mfma_common.h), the register-tiled time kernels in their float32 instantiation. LAVILA_F32_GENERIC=1
// in the environment (or lvl_debug_f32_generic) sends float32 calls to the generic kernels instead (A/B in the tests).
#include <stdlib.h>

#include "common.h"

int lvl_generic_divided_fwd(const void* qkv, void* out, float* lse, int B, int F, int N, int H, int mode, int dtype,
                            hipStream_t st, bool do_groups, bool do_cls);
int lvl_generic_divided_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                            float* ws, int B, int F, int N, int H, int mode, int dtype, hipStream_t st);
int lvl_generic_causal_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dtype, hipStream_t st);
int lvl_generic_causal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                           float* ws, int B, int L, int H, int dtype, hipStream_t st);
bool lvl_space_mfma_supported(int F, int N, int dtype);
int lvl_space_mfma_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, int dtype,
                       hipStream_t st);
// key-tiled streaming kernels for large space groups (attn_space_stream.hip)
bool lvl_space_stream_wanted(int F, int N, int dtype);
int lvl_space_stream_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, int dtype,
                         hipStream_t st);
int lvl_space_stream_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                         int B, int F, int N, int H, int dtype, hipStream_t st);
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
int lvl_text_mfma_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dtype, hipStream_t st);
bool lvl_text_mfma_bwd_supported(int L, int dtype);
int lvl_text_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int L, int H, int dtype, hipStream_t st);
bool lvl_space_mfma_bwd_supported(int F, int N, int dtype);
int lvl_space_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                       int B, int F, int N, int H, int dtype, hipStream_t st);
bool lvl_time_fast_bwd_supported(int F, int N, int H);
int lvl_time_fast_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int F, int N, int H, int dtype, hipStream_t st);
int lvl_time_fast_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, int dtype,
                      hipStream_t st);

// float32 calls take the fast kernels unless the environment (LAVILA_F32_GENERIC=1, read once) or the test hook
// lvl_debug_f32_generic asks for the generic ones
static std::atomic<int> g_f32_generic{-1};
static bool f32_fast() {
  int v = g_f32_generic.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("LAVILA_F32_GENERIC");
    v = (e && e[0] == '1') ? 1 : 0;
    g_f32_generic.store(v, std::memory_order_relaxed);
  }
  return v == 0;
}
// number of C-ABI attention calls that were served by the shape-generic kernels (tests assert 0 on the paths that must
// run on the fast kernels); reset != 0 clears it after reading
static std::atomic<int> g_generic_calls{0};
extern "C" int lvl_debug_generic_attention_calls(int reset) {
  return reset ? g_generic_calls.exchange(0) : g_generic_calls.load();
}
extern "C" int lvl_debug_f32_generic(int on) {
  g_f32_generic.store(on ? 1 : 0, std::memory_order_relaxed);
  return LVL_OK;
}
static bool fast_dtype(int dtype) { return dtype == LVL_BF16 || (dtype == LVL_F32 && f32_fast()); }

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
  if (fast_dtype(dtype) && mode == LVL_ATTN_SPACE && F <= 64 && lvl_space_stream_wanted(F, N, dtype))
    return lvl_space_stream_fwd(qkv, out, lse, ws, B, F, N, H, dtype, (hipStream_t)stream);
  if (fast_dtype(dtype) && mode == LVL_ATTN_SPACE && lvl_space_mfma_supported(F, N, dtype))
    return lvl_space_mfma_fwd(qkv, out, lse, ws, B, F, N, H, dtype, (hipStream_t)stream);
  if (dtype == LVL_BF16 && mode == LVL_ATTN_TIME && time_use_mfma(F, N, H))
    return lvl_time_mfma_fwd(qkv, out, lse, ws, B, F, N, H, (hipStream_t)stream);
  if (fast_dtype(dtype) && mode == LVL_ATTN_TIME && lvl_time_fast_supported(F, N, H))
    return lvl_time_fast_fwd(qkv, out, lse, ws, B, F, N, H, dtype, (hipStream_t)stream);
  g_generic_calls.fetch_add(1, std::memory_order_relaxed);
  return lvl_generic_divided_fwd(qkv, out, lse, B, F, N, H, mode, dtype, (hipStream_t)stream, true, true);
}

extern "C" int lvl_divided_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                    float* ws, int B, int F, int N, int H, int mode, int dtype, void* stream) {
  if (int rc = check_divided("divided_attn_bwd", qkv, out, B, F, N, H, mode, dtype)) return rc;
  LVL_REQUIRE(dout && lse && dqkv && ws, "divided_attn_bwd: null pointer");
  LVL_REQUIRE(lvl_aligned16(dout) && lvl_aligned16(dqkv), "divided_attn_bwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  if (fast_dtype(dtype) && mode == LVL_ATTN_SPACE && F <= 64 && lvl_space_stream_wanted(F, N, dtype))
    return lvl_space_stream_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, dtype, (hipStream_t)stream);
  if (fast_dtype(dtype) && mode == LVL_ATTN_SPACE && lvl_space_mfma_bwd_supported(F, N, dtype))
    return lvl_space_mfma_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, dtype, (hipStream_t)stream);
  if (dtype == LVL_BF16 && mode == LVL_ATTN_TIME && time_use_mfma(F, N, H))
    return lvl_time_mfma_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, (hipStream_t)stream);
  if (fast_dtype(dtype) && mode == LVL_ATTN_TIME && lvl_time_fast_bwd_supported(F, N, H))
    return lvl_time_fast_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, dtype, (hipStream_t)stream);
  g_generic_calls.fetch_add(1, std::memory_order_relaxed);
  return lvl_generic_divided_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, mode, dtype, (hipStream_t)stream);
}

extern "C" int lvl_causal_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dtype,
                                   void* stream) {
  LVL_REQUIRE(qkv && out && lse, "causal_attn_fwd: null pointer");
  LVL_REQUIRE(B >= 0 && L > 0 && H > 0, "causal_attn_fwd: bad shape B=%d L=%d H=%d", B, L, H);
  LVL_REQUIRE(lvl_aligned16(qkv) && lvl_aligned16(out), "causal_attn_fwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  if (fast_dtype(dtype) && lvl_text_mfma_supported(L))
    return lvl_text_mfma_fwd(qkv, out, lse, B, L, H, dtype, (hipStream_t)stream);
  g_generic_calls.fetch_add(1, std::memory_order_relaxed);
  return lvl_generic_causal_fwd(qkv, out, lse, B, L, H, dtype, (hipStream_t)stream);
}

extern "C" int lvl_causal_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                   float* ws, int B, int L, int H, int dtype, void* stream) {
  LVL_REQUIRE(qkv && out && dout && lse && dqkv && ws, "causal_attn_bwd: null pointer");
  LVL_REQUIRE(B >= 0 && L > 0 && H > 0, "causal_attn_bwd: bad shape B=%d L=%d H=%d", B, L, H);
  LVL_REQUIRE(lvl_aligned16(qkv) && lvl_aligned16(out) && lvl_aligned16(dout) && lvl_aligned16(dqkv),
              "causal_attn_bwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  if (fast_dtype(dtype) && lvl_text_mfma_bwd_supported(L, dtype))
    return lvl_text_mfma_bwd(qkv, out, dout, lse, dqkv, ws, B, L, H, dtype, (hipStream_t)stream);
  g_generic_calls.fetch_add(1, std::memory_order_relaxed);
  return lvl_generic_causal_bwd(qkv, out, dout, lse, dqkv, ws, B, L, H, dtype, (hipStream_t)stream);
}

// 1 if a bf16 call of this shape runs on the MFMA / register-tiled kernels (forward AND backward), 0 if it lands on
// the shape-generic kernels of attn_generic.hip (correct, latency-bound). Host-side query, no device work.
extern "C" int lvl_attention_fast_path(int mode, int F, int N, int H) {
  if (mode == LVL_ATTN_SPACE)
    return F <= 64 && (lvl_space_stream_wanted(F, N, LVL_BF16) ||
                       (lvl_space_mfma_supported(F, N, LVL_BF16) && lvl_space_mfma_bwd_supported(F, N, LVL_BF16)));
  if (mode == LVL_ATTN_TIME)
    return lvl_time_mfma_supported(F, N, H) || (lvl_time_fast_supported(F, N, H) && lvl_time_fast_bwd_supported(F, N, H));
  if (mode == LVL_ATTN_CAUSAL) return lvl_text_mfma_supported(N) && lvl_text_mfma_bwd_supported(N, LVL_BF16);
  return 0;
}

// the same question for float32 tensors (the parity configuration): 1 = the f32-class instantiations of the fast
// kernels (split-operand MFMA / float32 register-tiled), 0 = the shape-generic kernels
extern "C" int lvl_attention_fast_path_f32(int mode, int F, int N, int H) {
  if (!f32_fast()) return 0;
  if (mode == LVL_ATTN_SPACE)
    return F <= 64 && (lvl_space_stream_wanted(F, N, LVL_F32) ||
                       (lvl_space_mfma_supported(F, N, LVL_F32) && lvl_space_mfma_bwd_supported(F, N, LVL_F32)));
  if (mode == LVL_ATTN_TIME) return lvl_time_fast_supported(F, N, H) && lvl_time_fast_bwd_supported(F, N, H);
  if (mode == LVL_ATTN_CAUSAL) return lvl_text_mfma_supported(N) && lvl_text_mfma_bwd_supported(N, LVL_F32);
  return 0;
}
