// Contrastive (InfoNCE) head in slab form, gfx950: shape-generic VALU kernels (any E % 8 == 0) and the backward.
// The forward for E in {64,128,256,512} runs on the matrix cores (clip_loss_mfma.hip).
//
// Each rank owns B rows of both logit matrices: direction 0 = logits_per_image rows
// (scale*img_local) @ txt_all^T, direction 1 = logits_per_text rows (scale*txt_local) @ img_all^T.
// One workgroup per (row, direction): the [B,G] logits slab is never written to HBM (optional debug
// output aside); the row statistics (log-sum-exp, diagonal, expectation, argmax) come out of one
// streaming pass, and backward recomputes the logits from the gathered embeddings + gathered LSEs,
// so no gradient collective is needed.
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ float dot_row(const T* __restrict__ b, const float* __restrict__ a_s, int E) {
  float acc = 0.f;
  for (int e = 0; e < E; e += 8) {
    float v[8];
    Elem<T>::load8(b + e, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(a_s[e + j], v[j], acc);
  }
  return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void clip_fwd_kernel(const T* __restrict__ img_all, const T* __restrict__ txt_all,
                                                       const float* __restrict__ scale_p, int B, int G, int E, int row0,
                                                       float* __restrict__ stats, int32_t* __restrict__ argmax,
                                                       float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // a_s[E] then reduction scratch [4*5]
  float* a_s = smem;
  float* red = smem + E;
  const int i = blockIdx.x, dir = blockIdx.y, gi = row0 + i;
  const T* A = dir == 0 ? img_all : txt_all;
  const T* Bm = dir == 0 ? txt_all : img_all;
  const float scale = *scale_p;
  for (int e = threadIdx.x; e < E; e += blockDim.x) a_s[e] = scale * Elem<T>::load(A + (int64_t)gi * E + e);
  __syncthreads();

  float m = -INFINITY, l = 0.f, ex = 0.f, best = -INFINITY, diag = 0.f;
  int bi = 0x7fffffff;
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    const float z = dot_row(Bm + (int64_t)j * E, a_s, E);
    if (logits) logits[((int64_t)dir * B + i) * G + j] = z;
    if (j == gi) diag = z;
    if (z > best) { best = z; bi = j; }
    const float mn = fmaxf(m, z);
    const float al = __expf(m - mn), p = __expf(z - mn);
    l = l * al + p;
    ex = ex * al + p * z;
    m = mn;
  }
  // wave combine
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float M = wave_max(m);
  const float sc = (m == -INFINITY) ? 0.f : __expf(m - M);
  l = wave_sum(l * sc);
  ex = wave_sum(ex * sc);
  diag = wave_sum(diag);
  const float wbest = wave_max(best);
  int cand = (best == wbest) ? bi : 0x7fffffff;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  if (lane == 0) {
    red[wave * 5 + 0] = M; red[wave * 5 + 1] = l; red[wave * 5 + 2] = ex; red[wave * 5 + 3] = wbest;
    red[wave * 5 + 4] = __int_as_float(cand);
    red[20 + wave] = diag;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float MM = -INFINITY, bb = -INFINITY, dd = 0.f;
    for (int w = 0; w < 4; ++w) { MM = fmaxf(MM, red[w * 5]); bb = fmaxf(bb, red[w * 5 + 3]); dd += red[20 + w]; }
    float ll = 0.f, ee = 0.f;
    int idx = 0x7fffffff;
    for (int w = 0; w < 4; ++w) {
      const float s2 = (red[w * 5] == -INFINITY) ? 0.f : __expf(red[w * 5] - MM);
      ll += red[w * 5 + 1] * s2;
      ee += red[w * 5 + 2] * s2;
      if (red[w * 5 + 3] == bb) idx = min(idx, __float_as_int(red[w * 5 + 4]));
    }
    float* st = stats + ((int64_t)dir * B + i) * 4;
    st[0] = MM + __logf(ll);     // log-sum-exp of the row
    st[1] = dd;                   // diagonal (target) logit
    st[2] = ee / ll;              // sum_j softmax_j * logit_j
    st[3] = bb;                   // max logit
    argmax[dir * B + i] = idx;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void clip_bwd_kernel(const T* __restrict__ img_all, const T* __restrict__ txt_all,
                                                       const float* __restrict__ lse_all,
                                                       const float* __restrict__ scale_p,
                                                       const float* __restrict__ upstream_p, float coef, int B,
                                                       int G, int E, int row0, int rows_only,
                                                       float* __restrict__ dimg, float* __restrict__ dtxt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // a_s[E], c[G]
  float* a_s = smem;
  float* c = smem + E;
  const int i = blockIdx.x, dir = blockIdx.y, gi = row0 + i;
  const T* A = dir == 0 ? img_all : txt_all;
  const T* Bm = dir == 0 ? txt_all : img_all;
  const float scale = *scale_p;
  for (int e = threadIdx.x; e < E; e += blockDim.x) a_s[e] = scale * Elem<T>::load(A + (int64_t)gi * E + e);
  __syncthreads();
  const float L_own = lse_all[(int64_t)dir * G + gi];
  const float* L_other = lse_all + (int64_t)(1 - dir) * G;
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    const float z = dot_row(Bm + (int64_t)j * E, a_s, E);
    // rows_only (CLIPLoss(local_loss=True) without gather_with_grad, loss.py:86-88): the gathered partner rows carry
    // no gradient, so only this row's own cross-entropy contributes
    c[j] = rows_only ? __expf(z - L_own) - (j == gi ? 1.f : 0.f)
                     : __expf(z - L_own) + __expf(z - L_other[j]) - (j == gi ? 2.f : 0.f);
  }
  __syncthreads();
  float* dst = (dir == 0 ? dimg : dtxt) + (int64_t)i * E;
  const float k = coef * scale * (upstream_p ? *upstream_p : 1.0f);
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < G; ++j) acc = fmaf(c[j], Elem<T>::load(Bm + (int64_t)j * E + e), acc);
    dst[e] = k * acc;
  }
}

}  // namespace

bool lvl_clip_mfma_supported(int E);
int lvl_clip_fwd_mfma(const void* img_all, const void* txt_all, const float* scale, int B, int G, int E, int row0,
                      float* stats, int32_t* argmax, float* logits, int dtype, hipStream_t st);

extern "C" int lvl_clip_loss_fwd(const void* img_all, const void* txt_all, const float* scale, int B, int G, int E, int row0,
                                 float* stats, int32_t* argmax, float* logits, int dtype, void* stream) {
  LVL_REQUIRE(img_all && txt_all && scale && stats && argmax, "clip_loss_fwd: null pointer");
  LVL_REQUIRE(B >= 0 && G > 0 && E > 0 && E % 8 == 0 && row0 >= 0 && row0 + B <= G,
              "clip_loss_fwd: bad shape B=%d G=%d E=%d row0=%d", B, G, E, row0);
  LVL_REQUIRE(lvl_aligned16(img_all) && lvl_aligned16(txt_all), "clip_loss_fwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  if (lvl_clip_mfma_supported(E) && (dtype == LVL_F32 || dtype == LVL_BF16))
    return lvl_clip_fwd_mfma(img_all, txt_all, scale, B, G, E, row0, stats, argmax, logits, dtype, (hipStream_t)stream);
  const size_t shmem = (size_t)(E + 32) * sizeof(float);
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((clip_fwd_kernel<T>), dim3(B, 2), dim3(256), shmem, (hipStream_t)stream,
                                               (const T*)img_all, (const T*)txt_all, scale, B, G, E, row0, stats,
                                               argmax, logits));
  LVL_CHECK_LAUNCH("clip_loss_fwd");
  return LVL_OK;
}

extern "C" int lvl_clip_loss_bwd(const void* img_all, const void* txt_all, const float* lse_all, const float* scale,
                                 const float* upstream, float coef, int B, int G, int E, int row0, int rows_only,
                                 float* dimg, float* dtxt, int dtype, void* stream) {
  LVL_REQUIRE(img_all && txt_all && lse_all && scale && dimg && dtxt, "clip_loss_bwd: null pointer");
  LVL_REQUIRE(B >= 0 && G > 0 && E > 0 && E % 8 == 0 && row0 >= 0 && row0 + B <= G,
              "clip_loss_bwd: bad shape B=%d G=%d E=%d row0=%d", B, G, E, row0);
  LVL_REQUIRE((size_t)(E + G) * sizeof(float) <= 150 * 1024, "clip_loss_bwd: G=%d too large for the LDS-resident row", G);
  LVL_REQUIRE(lvl_aligned16(img_all) && lvl_aligned16(txt_all), "clip_loss_bwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  const size_t shmem = (size_t)(E + G) * sizeof(float);
  LVL_DISPATCH_DTYPE(dtype, {
    if (shmem > 64 * 1024)
      if (int rc = lvl_allow_lds<clip_bwd_kernel<T>>()) return rc;
    hipLaunchKernelGGL((clip_bwd_kernel<T>), dim3(B, 2), dim3(256), shmem, (hipStream_t)stream, (const T*)img_all,
                       (const T*)txt_all, lse_all, scale, upstream, coef, B, G, E, row0, rows_only, dimg, dtxt);
  });
  LVL_CHECK_LAUNCH("clip_loss_bwd");
  return LVL_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// SSLCLIPLoss (loss.py:121-217): per-pair temperature. scale(i,j) = scales[ind[i] + ind[j]] with
// scales = {pseudo, sqrt(pseudo*real), real}; logits = scale(i,j) * (a_i . b_j). Same slab structure; the row
// statistics additionally keep the three bucket expectations sum_{j in bucket k} p_j z_j (z = unscaled dot) that
// give d(loss)/d(scales[k]) without a second pass.
// ---------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void ssl_fwd_kernel(const T* __restrict__ img_all, const T* __restrict__ txt_all,
                                                      const int32_t* __restrict__ ind,
                                                      const float* __restrict__ scales, int B, int G, int E, int row0,
                                                      float* __restrict__ stats, int32_t* __restrict__ argmax,
                                                      float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // a_s[E], red[4][8]
  float* a_s = smem;
  float* red = smem + E;
  const int i = blockIdx.x, dir = blockIdx.y, gi = row0 + i;
  const T* A = dir == 0 ? img_all : txt_all;
  const T* Bm = dir == 0 ? txt_all : img_all;
  for (int e = threadIdx.x; e < E; e += blockDim.x) a_s[e] = Elem<T>::load(A + (int64_t)gi * E + e);
  __syncthreads();
  const int ind_i = ind[gi];
  const float sc0 = scales[ind_i], sc1 = scales[ind_i + 1];          // partner indicator 0 / 1

  float m = -INFINITY, l = 0.f, ex[3] = {0.f, 0.f, 0.f}, best = -INFINITY, diag_z = 0.f, diag_l = 0.f;
  int bi = 0x7fffffff;
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    const float zr = dot_row(Bm + (int64_t)j * E, a_s, E);
    const int bucket = ind_i + ind[j];
    const float z = zr * (ind[j] ? sc1 : sc0);
    if (logits) logits[((int64_t)dir * B + i) * G + j] = z;
    if (j == gi) { diag_z = zr; diag_l = z; }
    if (z > best) { best = z; bi = j; }
    const float mn = fmaxf(m, z);
    const float al = __expf(m - mn), p = __expf(z - mn);
    l = l * al + p;
#pragma unroll
    for (int k = 0; k < 3; ++k) ex[k] = ex[k] * al + (bucket == k ? p * zr : 0.f);
    m = mn;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float M = wave_max(m);
  const float sc = (m == -INFINITY) ? 0.f : __expf(m - M);
  l = wave_sum(l * sc);
#pragma unroll
  for (int k = 0; k < 3; ++k) ex[k] = wave_sum(ex[k] * sc);
  diag_z = wave_sum(diag_z);
  diag_l = wave_sum(diag_l);
  const float wbest = wave_max(best);
  int cand = (best == wbest) ? bi : 0x7fffffff;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  if (lane == 0) {
    float* r = red + wave * 9;
    r[0] = M; r[1] = l; r[2] = ex[0]; r[3] = ex[1]; r[4] = ex[2]; r[5] = wbest; r[6] = __int_as_float(cand);
    r[7] = diag_z; r[8] = diag_l;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float MM = -INFINITY, bb = -INFINITY, dz = 0.f, dl = 0.f;
    for (int w = 0; w < 4; ++w) { MM = fmaxf(MM, red[w * 9]); bb = fmaxf(bb, red[w * 9 + 5]); dz += red[w * 9 + 7]; dl += red[w * 9 + 8]; }
    float ll = 0.f, ee[3] = {0.f, 0.f, 0.f};
    int idx = 0x7fffffff;
    for (int w = 0; w < 4; ++w) {
      const float s2 = (red[w * 9] == -INFINITY) ? 0.f : __expf(red[w * 9] - MM);
      ll += red[w * 9 + 1] * s2;
      for (int k = 0; k < 3; ++k) ee[k] += red[w * 9 + 2 + k] * s2;
      if (red[w * 9 + 5] == bb) idx = min(idx, __float_as_int(red[w * 9 + 6]));
    }
    float* st = stats + ((int64_t)dir * B + i) * 8;
    st[0] = MM + __logf(ll);           // log-sum-exp of the row
    st[1] = dl;                         // diagonal (target) logit
    st[2] = ee[0] / ll; st[3] = ee[1] / ll; st[4] = ee[2] / ll;     // bucket expectations of the unscaled dot
    st[5] = dz;                         // diagonal unscaled dot
    st[6] = bb;                         // max logit
    st[7] = 0.f;
    argmax[dir * B + i] = idx;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ssl_bwd_kernel(const T* __restrict__ img_all, const T* __restrict__ txt_all,
                                                      const int32_t* __restrict__ ind,
                                                      const float* __restrict__ lse_all,
                                                      const float* __restrict__ scales,
                                                      const float* __restrict__ upstream_p, float coef, int B, int G,
                                                      int E, int row0, float* __restrict__ dimg,
                                                      float* __restrict__ dtxt) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // a_s[E], c[G]
  float* a_s = smem;
  float* c = smem + E;
  const int i = blockIdx.x, dir = blockIdx.y, gi = row0 + i;
  const T* A = dir == 0 ? img_all : txt_all;
  const T* Bm = dir == 0 ? txt_all : img_all;
  for (int e = threadIdx.x; e < E; e += blockDim.x) a_s[e] = Elem<T>::load(A + (int64_t)gi * E + e);
  __syncthreads();
  const int ind_i = ind[gi];
  const float sc0 = scales[ind_i], sc1 = scales[ind_i + 1];
  const float L_own = lse_all[(int64_t)dir * G + gi];
  const float* L_other = lse_all + (int64_t)(1 - dir) * G;
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    const float s = ind[j] ? sc1 : sc0;
    const float z = dot_row(Bm + (int64_t)j * E, a_s, E) * s;
    c[j] = s * (__expf(z - L_own) + __expf(z - L_other[j]) - (j == gi ? 2.f : 0.f));
  }
  __syncthreads();
  float* dst = (dir == 0 ? dimg : dtxt) + (int64_t)i * E;
  const float k = coef * (upstream_p ? *upstream_p : 1.0f);
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < G; ++j) acc = fmaf(c[j], Elem<T>::load(Bm + (int64_t)j * E + e), acc);
    dst[e] = k * acc;
  }
}

}  // namespace

extern "C" int lvl_ssl_clip_loss_fwd(const void* img_all, const void* txt_all, const int32_t* ind_all,
                                     const float* scales3, int B, int G, int E, int row0, float* stats,
                                     int32_t* argmax, float* logits, int dtype, void* stream) {
  LVL_REQUIRE(img_all && txt_all && ind_all && scales3 && stats && argmax, "ssl_clip_loss_fwd: null pointer");
  LVL_REQUIRE(B >= 0 && G > 0 && E > 0 && E % 8 == 0 && row0 >= 0 && row0 + B <= G,
              "ssl_clip_loss_fwd: bad shape B=%d G=%d E=%d row0=%d", B, G, E, row0);
  LVL_REQUIRE(lvl_aligned16(img_all) && lvl_aligned16(txt_all), "ssl_clip_loss_fwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  const size_t shmem = (size_t)(E + 40) * sizeof(float);
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ssl_fwd_kernel<T>), dim3(B, 2), dim3(256), shmem, (hipStream_t)stream,
                                               (const T*)img_all, (const T*)txt_all, ind_all, scales3, B, G, E, row0,
                                               stats, argmax, logits));
  LVL_CHECK_LAUNCH("ssl_clip_loss_fwd");
  return LVL_OK;
}

extern "C" int lvl_ssl_clip_loss_bwd(const void* img_all, const void* txt_all, const int32_t* ind_all,
                                     const float* lse_all, const float* scales3, const float* upstream, float coef,
                                     int B, int G, int E, int row0, float* dimg, float* dtxt, int dtype,
                                     void* stream) {
  LVL_REQUIRE(img_all && txt_all && ind_all && lse_all && scales3 && dimg && dtxt, "ssl_clip_loss_bwd: null pointer");
  LVL_REQUIRE(B >= 0 && G > 0 && E > 0 && E % 8 == 0 && row0 >= 0 && row0 + B <= G,
              "ssl_clip_loss_bwd: bad shape B=%d G=%d E=%d row0=%d", B, G, E, row0);
  LVL_REQUIRE((size_t)(E + G) * sizeof(float) <= 150 * 1024, "ssl_clip_loss_bwd: G=%d too large for the LDS-resident row", G);
  LVL_REQUIRE(lvl_aligned16(img_all) && lvl_aligned16(txt_all), "ssl_clip_loss_bwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  const size_t shmem = (size_t)(E + G) * sizeof(float);
  LVL_DISPATCH_DTYPE(dtype, {
    if (shmem > 64 * 1024)
      if (int rc = lvl_allow_lds<ssl_bwd_kernel<T>>()) return rc;
    hipLaunchKernelGGL((ssl_bwd_kernel<T>), dim3(B, 2), dim3(256), shmem, (hipStream_t)stream, (const T*)img_all,
                       (const T*)txt_all, ind_all, lse_all, scales3, upstream, coef, B, G, E, row0, dimg, dtxt);
  });
  LVL_CHECK_LAUNCH("ssl_clip_loss_bwd");
  return LVL_OK;
}
