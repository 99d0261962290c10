// Shape-generic grouped attention (forward + backward), f32 arithmetic, any element type.
//
// This is the always-correct path of the divided space-time attention and of the causal text
// attention: it serves the f32 parity configuration and every shape the MFMA kernels do not cover.
// One 64-lane wave owns one query row (forward / dq pass) or one key row (dk,dv pass):
//   * score phase: lane j holds key j of a 64-key chunk and computes q.k_j from its own row
//     (q replicated in registers) -> online softmax across chunks (no score buffer, any key count)
//   * value phase: lane d holds output channel d; p_j is broadcast with v_readlane
// No atomics, no cross-wave communication: results are deterministic.
#include "common.h"

namespace {

struct AttnDesc {
  int64_t qkv_bstride, qkv_tstride;   // elements between samples / tokens of the packed qkv tensor
  int64_t o_bstride, o_tstride;       // same for out / dout
  int H, T, groups;                   // heads, tokens per sample (lse is [B,H,T]), groups per sample
  int nq, q_first, q_gstride, q_istride;   // query i of group g is token q_first + g*q_gstride + i*q_istride
  int nk, k_first, k_gstride, k_istride;   // same for the group's keys (excluding the cls key)
  int has_cls;                        // key 0 of every group is token 0
  int causal;                         // key j visible to query i iff j <= i (needs has_cls == 0)
  float scale;
};

__device__ __forceinline__ int key_token(const AttnDesc& d, int g, int j) {
  if (d.has_cls) {
    if (j == 0) return 0;
    j -= 1;
  }
  return d.k_first + g * d.k_gstride + j * d.k_istride;
}

template <typename T>
__device__ __forceinline__ void load_row64(const T* p, float (&r)[64]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float v[8];
    Elem<T>::load8(p + c * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[c * 8 + j] = v[j];
  }
}

template <typename T>
__device__ __forceinline__ float dot_row64(const T* p, const float (&r)[64]) {
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float v[8];
    Elem<T>::load8(p + c * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(v[j], r[c * 8 + j], acc);
  }
  return acc;
}

__device__ __forceinline__ float bcast_lane(float v, int lane_uniform) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform));
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                       float* __restrict__ lse, AttnDesc d, int64_t total_q) {
  const int lane = threadIdx.x & 63;
  const int64_t qid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qid >= total_q) return;
  const int i = (int)(qid % d.nq);
  const int g = (int)((qid / d.nq) % d.groups);
  const int h = (int)((qid / ((int64_t)d.nq * d.groups)) % d.H);
  const int b = (int)(qid / ((int64_t)d.nq * d.groups * d.H));
  const int D = d.H * 64;
  const int tq = d.q_first + g * d.q_gstride + i * d.q_istride;
  const T* base = qkv + (int64_t)b * d.qkv_bstride + h * 64;

  float q[64];
  load_row64(base + (int64_t)tq * d.qkv_tstride, q);
#pragma unroll
  for (int c = 0; c < 64; ++c) q[c] *= d.scale;

  int nkt = d.has_cls + d.nk;
  if (d.causal && nkt > i + 1) nkt = i + 1;

  float m = -INFINITY, l = 0.f, acc = 0.f;
  for (int j0 = 0; j0 < nkt; j0 += 64) {
    const int j = j0 + lane;
    float s = -INFINITY;
    if (j < nkt) s = dot_row64(base + (int64_t)key_token(d, g, j) * d.qkv_tstride + D, q);
    const float m_new = fmaxf(m, wave_max(s));
    const float alpha = __expf(m - m_new);
    const float p = (j < nkt) ? __expf(s - m_new) : 0.f;
    l = l * alpha + wave_sum(p);
    acc *= alpha;
    const int cnt = (nkt - j0 < 64) ? (nkt - j0) : 64;
    for (int jj = 0; jj < cnt; ++jj) {
      const float pj = bcast_lane(p, jj);
      const T* vp = base + (int64_t)key_token(d, g, j0 + jj) * d.qkv_tstride + 2 * D;
      acc = fmaf(pj, Elem<T>::load(vp + lane), acc);
    }
    m = m_new;
  }
  Elem<T>::store(out + (int64_t)b * d.o_bstride + (int64_t)tq * d.o_tstride + h * 64 + lane, acc / l);
  if (lane == 0) lse[((int64_t)b * d.H + h) * d.T + tq] = m + __logf(l);
}

// dq pass: wave per query. Also writes delta_i = dout_i . out_i for the dk/dv pass.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                          const T* __restrict__ dout, const float* __restrict__ lse,
                                                          T* __restrict__ dqkv, float* __restrict__ delta,
                                                          AttnDesc d, int64_t total_q) {
  const int lane = threadIdx.x & 63;
  const int64_t qid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qid >= total_q) return;
  const int i = (int)(qid % d.nq);
  const int g = (int)((qid / d.nq) % d.groups);
  const int h = (int)((qid / ((int64_t)d.nq * d.groups)) % d.H);
  const int b = (int)(qid / ((int64_t)d.nq * d.groups * d.H));
  const int D = d.H * 64;
  const int tq = d.q_first + g * d.q_gstride + i * d.q_istride;
  const T* base = qkv + (int64_t)b * d.qkv_bstride + h * 64;
  const int64_t orow = (int64_t)b * d.o_bstride + (int64_t)tq * d.o_tstride + h * 64;

  float q[64], go[64];
  load_row64(base + (int64_t)tq * d.qkv_tstride, q);
  load_row64(dout + orow, go);
#pragma unroll
  for (int c = 0; c < 64; ++c) q[c] *= d.scale;
  const float dl = wave_sum(Elem<T>::load(dout + orow + lane) * Elem<T>::load(out + orow + lane));
  const int64_t srow = ((int64_t)b * d.H + h) * d.T + tq;
  const float L = lse[srow];
  if (lane == 0) delta[srow] = dl;

  int nkt = d.has_cls + d.nk;
  if (d.causal && nkt > i + 1) nkt = i + 1;
  float acc = 0.f;
  for (int j0 = 0; j0 < nkt; j0 += 64) {
    const int j = j0 + lane;
    float ds = 0.f;
    if (j < nkt) {
      const T* kp = base + (int64_t)key_token(d, g, j) * d.qkv_tstride + D;
      const float s = dot_row64(kp, q);
      const float dp = dot_row64(kp + D, go);
      ds = __expf(s - L) * (dp - dl);
    }
    const int cnt = (nkt - j0 < 64) ? (nkt - j0) : 64;
    for (int jj = 0; jj < cnt; ++jj) {
      const float dsj = bcast_lane(ds, jj);
      const T* kp = base + (int64_t)key_token(d, g, j0 + jj) * d.qkv_tstride + D;
      acc = fmaf(dsj, Elem<T>::load(kp + lane), acc);
    }
  }
  T* dq = dqkv + (int64_t)b * d.qkv_bstride + (int64_t)tq * d.qkv_tstride + h * 64 + lane;
  Elem<T>::store(dq, acc * d.scale);
}

// dk/dv pass: wave per key token. Key 0 (cls) of a has_cls descriptor is visited by the queries of
// every group; any other key only by its own group's queries.
template <typename T, bool ACCUM>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta, T* __restrict__ dqkv,
                                                           AttnDesc d, int64_t total_k) {
  const int lane = threadIdx.x & 63;
  const int64_t kid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (kid >= total_k) return;
  const int nkeys = d.has_cls + d.groups * d.nk;      // distinct key tokens per (b,h)
  const int kt = (int)(kid % nkeys);
  const int h = (int)((kid / nkeys) % d.H);
  const int b = (int)(kid / ((int64_t)nkeys * d.H));
  const int D = d.H * 64;
  int g = 0, jj_in_group = 0, tk = 0, nqs, all_groups = 0;
  if (d.has_cls && kt == 0) {
    all_groups = 1;
    nqs = d.groups * d.nq;
  } else {
    const int r = kt - d.has_cls;
    g = r / d.nk;
    jj_in_group = r - g * d.nk;
    tk = d.k_first + g * d.k_gstride + jj_in_group * d.k_istride;
    nqs = d.nq;
  }
  const T* base = qkv + (int64_t)b * d.qkv_bstride + h * 64;
  const T* dob = dout + (int64_t)b * d.o_bstride + h * 64;
  float k[64], v[64];
  load_row64(base + (int64_t)tk * d.qkv_tstride + D, k);
  load_row64(base + (int64_t)tk * d.qkv_tstride + 2 * D, v);
  const float* lrow = lse + ((int64_t)b * d.H + h) * d.T;
  const float* drow = delta + ((int64_t)b * d.H + h) * d.T;

  float dk = 0.f, dv = 0.f;
  for (int i0 = 0; i0 < nqs; i0 += 64) {
    const int qq = i0 + lane;
    float p = 0.f, ds = 0.f;
    if (qq < nqs) {
      const int gq = all_groups ? qq / d.nq : g;
      const int iq = all_groups ? qq - gq * d.nq : qq;
      const bool visible = !d.causal || iq >= jj_in_group;
      if (visible) {
        const int tq = d.q_first + gq * d.q_gstride + iq * d.q_istride;
        const float s = dot_row64(base + (int64_t)tq * d.qkv_tstride, k) * d.scale;
        const float dp = dot_row64(dob + (int64_t)tq * d.o_tstride, v);
        p = __expf(s - lrow[tq]);
        ds = p * (dp - drow[tq]);
      }
    }
    const int cnt = (nqs - i0 < 64) ? (nqs - i0) : 64;
    for (int ii = 0; ii < cnt; ++ii) {
      const float pi = bcast_lane(p, ii), dsi = bcast_lane(ds, ii);
      const int qq2 = i0 + ii;
      const int gq = all_groups ? qq2 / d.nq : g;
      const int iq = all_groups ? qq2 - gq * d.nq : qq2;
      const int tq = d.q_first + gq * d.q_gstride + iq * d.q_istride;
      dv = fmaf(pi, Elem<T>::load(dob + (int64_t)tq * d.o_tstride + lane), dv);
      dk = fmaf(dsi, Elem<T>::load(base + (int64_t)tq * d.qkv_tstride + lane), dk);
    }
  }
  dk *= d.scale;
  T* dkp = dqkv + (int64_t)b * d.qkv_bstride + (int64_t)tk * d.qkv_tstride + D + h * 64 + lane;
  T* dvp = dkp + D;
  if (ACCUM) {
    dk += Elem<T>::load(dkp);
    dv += Elem<T>::load(dvp);
  }
  Elem<T>::store(dkp, dk);
  Elem<T>::store(dvp, dv);
}

inline AttnDesc group_desc(int F, int N, int H, int mode) {
  AttnDesc d{};
  const int T = 1 + F * N, D = H * 64;
  d.qkv_tstride = 3 * D; d.qkv_bstride = (int64_t)T * 3 * D;
  d.o_tstride = D; d.o_bstride = (int64_t)T * D;
  d.H = H; d.T = T; d.has_cls = 1; d.causal = 0; d.scale = 0.125f;
  if (mode == LVL_ATTN_SPACE) {
    d.groups = F; d.nq = N; d.q_first = 1; d.q_gstride = N; d.q_istride = 1;
  } else {
    d.groups = N; d.nq = F; d.q_first = 1; d.q_gstride = 1; d.q_istride = N;
  }
  d.nk = d.nq; d.k_first = d.q_first; d.k_gstride = d.q_gstride; d.k_istride = d.q_istride;
  return d;
}

inline AttnDesc cls_row_desc(int F, int N, int H) {
  AttnDesc d = group_desc(F, N, H, LVL_ATTN_SPACE);
  d.groups = 1; d.has_cls = 0;
  d.nq = 1; d.q_first = 0; d.q_gstride = 0; d.q_istride = 0;
  d.nk = d.T; d.k_first = 0; d.k_gstride = 0; d.k_istride = 1;
  return d;
}

inline AttnDesc causal_desc(int L, int H) {
  AttnDesc d{};
  const int D = H * 64;
  d.qkv_tstride = 3 * D; d.qkv_bstride = (int64_t)L * 3 * D;
  d.o_tstride = D; d.o_bstride = (int64_t)L * D;
  d.H = H; d.T = L; d.groups = 1; d.has_cls = 0; d.causal = 1; d.scale = 0.125f;
  d.nq = L; d.q_first = 0; d.q_gstride = 0; d.q_istride = 1;
  d.nk = L; d.k_first = 0; d.k_gstride = 0; d.k_istride = 1;
  return d;
}

template <typename T>
int run_fwd(const void* qkv, void* out, float* lse, const AttnDesc& d, int B, hipStream_t st) {
  const int64_t total = (int64_t)B * d.H * d.groups * d.nq;
  if (total == 0) return LVL_OK;
  hipLaunchKernelGGL((attn_fwd_kernel<T>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, (const T*)qkv,
                     (T*)out, lse, d, total);
  LVL_CHECK_LAUNCH("attn_fwd_generic");
  return LVL_OK;
}

template <typename T>
int run_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
            const AttnDesc& d, int B, bool accum_dkv, hipStream_t st) {
  const int64_t total_q = (int64_t)B * d.H * d.groups * d.nq;
  if (total_q == 0) return LVL_OK;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), dim3((unsigned)((total_q + 3) / 4)), dim3(256), 0, st, (const T*)qkv,
                     (const T*)out, (const T*)dout, lse, (T*)dqkv, delta, d, total_q);
  LVL_CHECK_LAUNCH("attn_bwd_dq_generic");
  const int64_t total_k = (int64_t)B * d.H * (d.has_cls + d.groups * d.nk);
  if (accum_dkv)
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, true>), dim3((unsigned)((total_k + 3) / 4)), dim3(256), 0, st,
                       (const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv, d, total_k);
  else
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, false>), dim3((unsigned)((total_k + 3) / 4)), dim3(256), 0, st,
                       (const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv, d, total_k);
  LVL_CHECK_LAUNCH("attn_bwd_dkv_generic");
  return LVL_OK;
}

}  // namespace

// entry points used by the dispatcher in attention.hip
int lvl_generic_divided_fwd(const void* qkv, void* out, float* lse, int B, int F, int N, int H, int mode, int dtype,
                            hipStream_t st, bool do_groups, bool do_cls) {
  int rc = LVL_OK;
  if (do_groups) {
    const AttnDesc d = group_desc(F, N, H, mode);
    LVL_DISPATCH_DTYPE(dtype, rc = run_fwd<T>(qkv, out, lse, d, B, st));
    if (rc) return rc;
  }
  if (do_cls) {
    const AttnDesc c = cls_row_desc(F, N, H);
    LVL_DISPATCH_DTYPE(dtype, rc = run_fwd<T>(qkv, out, lse, c, B, st));
  }
  return rc;
}

int lvl_generic_divided_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                            float* ws, int B, int F, int N, int H, int mode, int dtype, hipStream_t st) {
  int rc = LVL_OK;
  const AttnDesc d = group_desc(F, N, H, mode);
  LVL_DISPATCH_DTYPE(dtype, rc = run_bwd<T>(qkv, out, dout, lse, dqkv, ws, d, B, false, st));
  if (rc) return rc;
  const AttnDesc c = cls_row_desc(F, N, H);
  LVL_DISPATCH_DTYPE(dtype, rc = run_bwd<T>(qkv, out, dout, lse, dqkv, ws, c, B, true, st));
  return rc;
}

int lvl_generic_causal_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dtype, hipStream_t st) {
  int rc = LVL_OK;
  const AttnDesc d = causal_desc(L, H);
  LVL_DISPATCH_DTYPE(dtype, rc = run_fwd<T>(qkv, out, lse, d, B, st));
  return rc;
}

int lvl_generic_causal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                           float* ws, int B, int L, int H, int dtype, hipStream_t st) {
  int rc = LVL_OK;
  const AttnDesc d = causal_desc(L, H);
  LVL_DISPATCH_DTYPE(dtype, rc = run_bwd<T>(qkv, out, dout, lse, dqkv, ws, d, B, false, st));
  return rc;
}
