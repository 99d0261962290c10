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
                       float* dq_part, int B, int F, int N, int H, int dtype, hipStream_t st);
int lvl_space_mfma_bwd_dq_part_rows(int B, int F, int N);
bool lvl_time_fast_bwd_supported(int F, int N, int H);
int lvl_time_fast_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      float* dq_part, int B, int F, int N, int H, int dtype, hipStream_t st);
int lvl_time_fast_bwd_dq_part_rows(int B, int F, int N, int H);
int lvl_colsum_mid_rows();
int lvl_launch_column_reduce(const float* part, int nparts, int width, int seg, float* mid, float* out0, float* out1,
                             float* out2, hipStream_t st);
int lvl_launch_column_reduce_tail(const float* part, int nparts, int width, int seg, float* mid, float* out0, float* out1,
                                  float* out2, float* zero_dst, const float* copy_src, float* copy_dst, int tail_n,
                                  hipStream_t st);
int lvl_qkv_bias_sources(const void* dqkv, const void* dout, float* dbias, float* ws, int64_t rows, int D, int dtype,
                         bool zero_k, const float* v_src, hipStream_t st);
int64_t lvl_qkv_bias_ws_floats(int D);
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

// which backward kernel family a call lands on, and how many rows of dq column-sum partials it can emit on the way
// (0: no rider in that family -- the bias gradient's q third is then reduced from dqkv afterwards)
enum { BWD_STREAM, BWD_SPACE_MFMA, BWD_TIME_MFMA, BWD_TIME_FAST, BWD_GENERIC };
static int bwd_family(int B, int F, int N, int H, int mode, int dtype, int* part_rows) {
  *part_rows = 0;
  if (fast_dtype(dtype) && mode == LVL_ATTN_SPACE && F <= 64 && lvl_space_stream_wanted(F, N, dtype)) return BWD_STREAM;
  if (fast_dtype(dtype) && mode == LVL_ATTN_SPACE && lvl_space_mfma_bwd_supported(F, N, dtype)) {
    *part_rows = lvl_space_mfma_bwd_dq_part_rows(B, F, N);
    return BWD_SPACE_MFMA;
  }
  if (dtype == LVL_BF16 && mode == LVL_ATTN_TIME && time_use_mfma(F, N, H)) return BWD_TIME_MFMA;
  if (fast_dtype(dtype) && mode == LVL_ATTN_TIME && lvl_time_fast_bwd_supported(F, N, H)) {
    *part_rows = lvl_time_fast_bwd_dq_part_rows(B, F, N, H);
    return BWD_TIME_FAST;
  }
  return BWD_GENERIC;
}

static int divided_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                       float* dq_part, int B, int F, int N, int H, int mode, int dtype, hipStream_t st) {
  int part_rows;
  switch (bwd_family(B, F, N, H, mode, dtype, &part_rows)) {
    case BWD_STREAM: return lvl_space_stream_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, dtype, st);
    case BWD_SPACE_MFMA: return lvl_space_mfma_bwd(qkv, out, dout, lse, dqkv, ws, dq_part, B, F, N, H, dtype, st);
    case BWD_TIME_MFMA: return lvl_time_mfma_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, st);
    case BWD_TIME_FAST: return lvl_time_fast_bwd(qkv, out, dout, lse, dqkv, ws, dq_part, B, F, N, H, dtype, st);
    default: break;
  }
  g_generic_calls.fetch_add(1, std::memory_order_relaxed);
  return lvl_generic_divided_bwd(qkv, out, dout, lse, dqkv, ws, B, F, N, H, mode, dtype, st);
}

extern "C" int lvl_divided_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                    float* ws, int B, int F, int N, int H, int mode, int dtype, void* stream) {
  if (int rc = check_divided("divided_attn_bwd", qkv, out, B, F, N, H, mode, dtype)) return rc;
  LVL_REQUIRE(dout && lse && dqkv && ws, "divided_attn_bwd: null pointer");
  LVL_REQUIRE(lvl_aligned16(dout) && lvl_aligned16(dqkv), "divided_attn_bwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  return divided_bwd(qkv, out, dout, lse, dqkv, ws, nullptr, B, F, N, H, mode, dtype, (hipStream_t)stream);
}

// ---- backward + the gradient of the qkv Linear's bias in one call (round 5) ---------------------------------------------
// d(bias) [3D] = column sums of dqkv over all B*T rows. Thirds:
//   k: exactly 0 (the scores do not change when one vector is added to every key of a sample);
//   v: column sums of dout (every softmax row sums to 1) -- handed over by the caller as `dout_colsum` when it already has
//      them (dout = dy . W_proj of the projection Linear, so sum_rows(dout) = sum_rows(dy) . W_proj = d(b_proj) . W_proj: a
//      vector-matrix product instead of a pass over dout, lvl_vec_mat_f32), else reduced here from dout;
//   q: partial column sums written by the backward kernel itself where the family has the rider (the LDS-resident fused
//      space kernel, the register-tiled time kernels: the dQ accumulators / dq rows are in registers there), else a pass
//      over the q third of dqkv.
// ws2: lvl_divided_attn_bwd_bias_ws floats.
extern "C" int64_t lvl_divided_attn_bwd_bias_ws(int B, int F, int N, int H, int mode, int dtype) {
  int part_rows = 0;
  if (B > 0 && F > 0 && N > 0 && H > 0) bwd_family(B, F, N, H, mode, dtype, &part_rows);
  const int64_t D = (int64_t)H * 64;
  const int64_t rider = ((int64_t)part_rows + lvl_colsum_mid_rows()) * D;
  const int64_t pass = lvl_qkv_bias_ws_floats((int)D);
  return rider > pass ? rider : pass;
}

extern "C" int lvl_divided_attn_bwd_bias(const void* qkv, const void* out, const void* dout, const float* lse,
                                         void* dqkv, float* ws, const float* dout_colsum, float* dbias, float* ws2,
                                         int B, int F, int N, int H, int mode, int dtype, void* stream) {
  if (int rc = check_divided("divided_attn_bwd_bias", qkv, out, B, F, N, H, mode, dtype)) return rc;
  LVL_REQUIRE(dout && lse && dqkv && ws && dbias && ws2, "divided_attn_bwd_bias: null pointer");
  LVL_REQUIRE(lvl_aligned16(dout) && lvl_aligned16(dqkv) && lvl_aligned16(dbias) && lvl_aligned16(ws2) &&
                  lvl_aligned16(dout_colsum), "divided_attn_bwd_bias: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int D = H * 64;
  const int64_t rows = (int64_t)B * (1 + (int64_t)F * N);
  if (B == 0) {
    return lvl_zero_f32(dbias, (size_t)3 * D, st);
  }
  int part_rows;
  bwd_family(B, F, N, H, mode, dtype, &part_rows);
  float* dq_part = part_rows > 0 ? ws2 : nullptr;
  if (int rc = divided_bwd(qkv, out, dout, lse, dqkv, ws, dq_part, B, F, N, H, mode, dtype, st)) return rc;
  // the zeros of the k third and a ready-made v third ride on the second stage of whichever reduction runs last
  if (dq_part) {
    float* v_dst = dbias + 2 * (size_t)D;
    if (int rc = lvl_launch_column_reduce_tail(dq_part, part_rows, D, D, dq_part + (size_t)part_rows * D, dbias, nullptr,
                                               nullptr, dbias + D, dout_colsum, dout_colsum ? v_dst : nullptr, D, st))
      return rc;
    if (dout_colsum) return LVL_OK;
    return lvl_qkv_bias_sources(nullptr, dout, dbias, ws2, rows, D, dtype, false, nullptr, st);      // v third from dout
  }
  return lvl_qkv_bias_sources(dqkv, dout_colsum ? nullptr : dout, dbias, ws2, rows, D, dtype, true, dout_colsum, st);
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
