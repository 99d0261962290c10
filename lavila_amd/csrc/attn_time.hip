// Time-mode divided attention forward + backward (bf16 or float32 in/out -- template E -- f32 arithmetic), gfx950.
//
// Per (sample b, location n, head h): F queries x (1 cls + F) keys, head dim 64 (timesformer.py:121-131
// with the '(b n) f d' grouping :302-303). 2.5 flop/B at F=4: purely HBM-bound, nothing for the matrix
// cores to do. 64/DPL lanes share one problem (DPL channels each), consecutive lane groups take consecutive
// heads, so a wave's load of "row f" covers whole 128-B head rows; K and V of the group stay packed in
// registers; every row of qkv is read once and every row of out / dqkv is written once. With DPL = 4 the
// kernels fit ~128 VGPRs: more waves per SIMD and every load of a location issued up front (the DPL = 8
// variant needed 256 VGPRs and ran at half the bandwidth).
// The CLS query (token 0) attends to ALL keys: each thread group folds its own F keys into a running
// flash-style partial (max, sum, acc) for its head while the rows are in registers; partials are merged per
// workgroup through LDS and across workgroups by cls_combine_kernel (attn_space_mfma.hip). In the backward
// the CLS row's rank-1 terms are folded into dk/dv, and d(cls q), d(cls k), d(cls v) -- which collect
// gradient from every location -- go through LDS + f32 atomics into a workspace (cls_grad_finalize_kernel).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int CLS_REC = 66;
constexpr float kLog2e = 1.4426950408889634f;

// E = element type in HBM: uint16_t (bf16 bits, the benched path) or float (the parity configuration: the same kernels,
// rows kept unpacked in registers)
template <typename E, int DPL> struct Vec;
template <int DPL> struct Vec<float, DPL> {
  struct alignas(DPL * 4 > 16 ? 16 : DPL * 4) type { float v[DPL]; };
  static __device__ __forceinline__ void unpack(const type& a, float (&v)[DPL]) {
#pragma unroll
    for (int i = 0; i < DPL; ++i) v[i] = a.v[i];
  }
  static __device__ __forceinline__ type pack(const float (&v)[DPL]) {
    type t;
#pragma unroll
    for (int i = 0; i < DPL; ++i) t.v[i] = v[i];
    return t;
  }
};
template <> struct Vec<uint16_t, 8> {
  using type = uint4;
  static __device__ __forceinline__ void unpack(const uint4& a, float (&v)[8]) {
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
    return make_uint4(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]), f32x2_to_bf16x2(v[4], v[5]),
                      f32x2_to_bf16x2(v[6], v[7]));
  }
};
template <> struct Vec<uint16_t, 4> {
  using type = uint2;
  static __device__ __forceinline__ void unpack(const uint2& a, float (&v)[4]) {
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  }
  static __device__ __forceinline__ uint2 pack(const float (&v)[4]) {
    return make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
  }
};

template <> struct Vec<uint16_t, 2> {
  using type = uint32_t;
  static __device__ __forceinline__ void unpack(const uint32_t& a, float (&v)[2]) {
    v[0] = __uint_as_float(a << 16); v[1] = __uint_as_float(a & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t pack(const float (&v)[2]) { return f32x2_to_bf16x2(v[0], v[1]); }
};

template <typename E, int DPL>
__device__ __forceinline__ float dotp(const float (&a)[DPL], const typename Vec<E, DPL>::type& b) {
  float v[DPL];
  Vec<E, DPL>::unpack(b, v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; ++i) s = fmaf(a[i], v[i], s);
  return s;
}

// sum over the 64/DPL lanes of one problem (an aligned 8-, 16- or 32-lane group): quad xor 1, xor 2, half-mirror,
// mirror are plain VALU DPP operands (no LDS crossbar round trips); only the 32-lane groups of DPL = 2 (many
// frames: 2 channels per lane keep the F key/value rows in registers) need one cross-row exchange
template <int DPL>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  if (DPL <= 4) v += dpp_move<0x140>(v);
  if (DPL == 2) v += __shfl_xor(v, 16, 64);
  return v;
}

// block = (64/DPL) * H * NPB threads; thread group = fixed head h, location slot n_sub
template <typename E, int F, int DPL>
__global__ __launch_bounds__((DPL == 2 ? 512 : 256), (DPL == 2 || sizeof(E) == 4 ? 1 : (F <= 4 ? 4 : 2))) void time_fwd_kernel(const E* __restrict__ qkv, E* __restrict__ out,
                                                       float* __restrict__ lse, float* __restrict__ cls_ws, int N,
                                                       int H, int NPB, int NCH, int NC) {
  using V = Vec<E, DPL>;
  using vec_t = typename V::type;
  constexpr int LPP = 64 / DPL;
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [NPB][H][LPP][2 + DPL]
  const int tid = threadIdx.x, dl = tid % LPP, grp = tid / LPP;
  const int h = grp % H, n_sub = grp / H;
  const int chunk = blockIdx.x % NC, b = blockIdx.x / NC;
  const int D = H * 64, T = 1 + F * N;
  const size_t ts = (size_t)3 * D;
  const E* base = qkv + (size_t)b * T * ts + h * 64 + dl * DPL;
  E* obase = out + (size_t)b * T * D + h * 64 + dl * DPL;
  float* lrow = lse + ((size_t)b * H + h) * T;

  const vec_t kc = *reinterpret_cast<const vec_t*>(base + D);           // cls key / value / query of this head
  const vec_t vc = *reinterpret_cast<const vec_t*>(base + 2 * D);
  float qc[DPL];
  V::unpack(*reinterpret_cast<const vec_t*>(base), qc);
#pragma unroll
  for (int i = 0; i < DPL; ++i) qc[i] *= 0.125f * kLog2e;              // scores in log2 units: exp2 only

  float cm = -INFINITY, cl = 0.f, ca[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) ca[i] = 0.f;
  if (chunk == 0 && n_sub == 0) {           // the cls key itself enters the CLS row exactly once per (b,h)
    cm = group_sum<DPL>(dotp<E, DPL>(qc, kc));
    cl = 1.f;
    V::unpack(vc, ca);
  }

  const int n_end = min(N, (chunk + 1) * NCH);
#pragma unroll 1
  for (int n = chunk * NCH + n_sub; n < n_end; n += NPB) {
    // every row of the location is requested here, and nothing below is a CONDITIONAL vector-memory instruction (the lse
    // store is issued by every lane of a group: one address) -- with a branch around a store in the loop the compiler's
    // wait insertion falls back to vmcnt(0) in front of every query, i.e. one store acknowledgement per query (round 6)
    vec_t kk[F], vv[F], qq[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const E* p = base + (size_t)(1 + f * N + n) * ts;
      qq[f] = *reinterpret_cast<const vec_t*>(p);
      kk[f] = *reinterpret_cast<const vec_t*>(p + D);
      vv[f] = *reinterpret_cast<const vec_t*>(p + 2 * D);
    }
    // CLS-query partial over this location's F keys
    {
      float s[F], mx = cm;
#pragma unroll
      for (int f = 0; f < F; ++f) { s[f] = group_sum<DPL>(dotp<E, DPL>(qc, kk[f])); mx = fmaxf(mx, s[f]); }
      const float al = __builtin_amdgcn_exp2f(cm - mx);
      cl *= al;
#pragma unroll
      for (int i = 0; i < DPL; ++i) ca[i] *= al;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float p = __builtin_amdgcn_exp2f(s[f] - mx);
        float v[DPL];
        V::unpack(vv[f], v);
        cl += p;
#pragma unroll
        for (int i = 0; i < DPL; ++i) ca[i] = fmaf(p, v[i], ca[i]);
      }
      cm = mx;
    }
    // the F patch queries of this location
#pragma unroll
    for (int fq = 0; fq < F; ++fq) {
      float q[DPL];
      V::unpack(qq[fq], q);
#pragma unroll
      for (int i = 0; i < DPL; ++i) q[i] *= 0.125f * kLog2e;
      float s[F + 1];
      s[0] = group_sum<DPL>(dotp<E, DPL>(q, kc));
      float mx = s[0];
#pragma unroll
      for (int f = 0; f < F; ++f) { s[f + 1] = group_sum<DPL>(dotp<E, DPL>(q, kk[f])); mx = fmaxf(mx, s[f + 1]); }
      float o[DPL], v[DPL];
      float p = __builtin_amdgcn_exp2f(s[0] - mx), l = p;
      V::unpack(vc, v);
#pragma unroll
      for (int i = 0; i < DPL; ++i) o[i] = p * v[i];
#pragma unroll
      for (int f = 0; f < F; ++f) {
        p = __builtin_amdgcn_exp2f(s[f + 1] - mx);
        l += p;
        V::unpack(vv[f], v);
#pragma unroll
        for (int i = 0; i < DPL; ++i) o[i] = fmaf(p, v[i], o[i]);
      }
      const float linv = __builtin_amdgcn_rcpf(l);
#pragma unroll
      for (int i = 0; i < DPL; ++i) o[i] *= linv;
      const int tok = 1 + fq * N + n;
      *reinterpret_cast<vec_t*>(obase + (size_t)tok * D) = V::pack(o);
      lrow[tok] = (mx + __log2f(l)) * (1.0f / kLog2e);          // all lanes of the group, the same value
    }
  }

  // merge the NPB location slots of each head, one record per (b, h, chunk); cm is in log2 units
  constexpr int RS = 2 + DPL;
  float* mine = smem + ((size_t)(n_sub * H + h) * LPP + dl) * RS;
  mine[0] = cm; mine[1] = cl;
#pragma unroll
  for (int i = 0; i < DPL; ++i) mine[2 + i] = ca[i];
  __syncthreads();
  if (n_sub == 0) {
    float M = -INFINITY;
    for (int s = 0; s < NPB; ++s) M = fmaxf(M, smem[((size_t)(s * H + h) * LPP + dl) * RS]);
    float Ls = 0.f, acc[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = 0.f;
    for (int s = 0; s < NPB; ++s) {
      const float* r = smem + ((size_t)(s * H + h) * LPP + dl) * RS;
      const float w = r[0] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(r[0] - M);
      Ls = fmaf(r[1], w, Ls);
#pragma unroll
      for (int i = 0; i < DPL; ++i) acc[i] = fmaf(r[2 + i], w, acc[i]);
    }
    float* rec = cls_ws + (((size_t)b * H + h) * NC + chunk) * CLS_REC;
    if (dl == 0) { rec[0] = M * (1.0f / kLog2e); rec[1] = Ls; }      // record max in natural-log units
#pragma unroll
    for (int i = 0; i < DPL; ++i) rec[2 + dl * DPL + i] = acc[i];
  }
}

// ---- backward ------------------------------------------------------------------------------------------------
// Everything of a (b, n, h) problem is thread-group local: the softmax is recomputed from the F+1 keys in
// registers (no saved statistics needed, and delta = sum_j P dP needs no O rows: `out` is only read for the cls
// row), dq/dk/dv rows of the patch tokens are written exactly once. HBM: 5 row reads + 3 row writes per token.
// RIDER (round 5): 0 = no bias-gradient rider (the kernel of rounds 1-4, instruction for instruction); 1 = dq column sums
// in DPL registers per thread at the kernel's usual occupancy; 2 = the same at one wave per SIMD less (the F <= 4 bf16
// instantiations sit AT their register budget: 13 spilled registers without the rider, 22 with it at 3 waves per SIMD,
// none at 2). Which one runs is a measured choice (lvl_debug_time_bwd_rider, profiles/r05_time_bwd_rider.txt).
constexpr int time_bwd_waves(int esize, int F, int DPL, int RIDER) {
  const int w = (DPL == 2 || esize == 4) ? 1 : (F <= 2 ? 4 : (F <= 4 ? 3 : 2));
  return (RIDER == 2 && w > 1) ? w - 1 : w;
}
template <typename E, int F, int DPL, int RIDER = 0>
__global__ __launch_bounds__((DPL == 2 ? 512 : 256), time_bwd_waves(sizeof(E), F, DPL, RIDER)) void time_bwd_kernel(const E* __restrict__ qkv, const E* __restrict__ out,
                                                       const E* __restrict__ dout, const float* __restrict__ lse,
                                                       E* __restrict__ dqkv, float* __restrict__ atom_ws,
                                                       float* __restrict__ dq_part, int N,
                                                       int H, int NPB, int NCH, int NC) {
  // dq_part (nullable; round 5): [B * NC, H * 64] f32 -- column sums of the dq rows this workgroup writes plus its share
  // of the cls query's dq: its part of the q third of d(qkv bias), accumulated in DPL registers per thread (the rows
  // are in registers anyway) instead of a pass over dqkv afterwards.
  using V = Vec<E, DPL>;
  using vec_t = typename V::type;
  constexpr int LPP = 64 / DPL;
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [NPB][H][LPP][4 * DPL]
  const int tid = threadIdx.x, dl = tid % LPP, grp = tid / LPP;
  const int h = grp % H, n_sub = grp / H;
  const int chunk = blockIdx.x % NC, b = blockIdx.x / NC;
  const int D = H * 64, T = 1 + F * N;
  const size_t ts = (size_t)3 * D;
  const E* base = qkv + (size_t)b * T * ts + h * 64 + dl * DPL;
  E* gbase = dqkv + (size_t)b * T * ts + h * 64 + dl * DPL;
  const E* obase = out + (size_t)b * T * D + h * 64 + dl * DPL;
  const E* dobase = dout + (size_t)b * T * D + h * 64 + dl * DPL;

  float qc[DPL], doc[DPL], kc[DPL], vc[DPL];
  V::unpack(*reinterpret_cast<const vec_t*>(base), qc);              // raw cls query
  V::unpack(*reinterpret_cast<const vec_t*>(dobase), doc);           // d out of the cls row
  V::unpack(*reinterpret_cast<const vec_t*>(base + D), kc);
  V::unpack(*reinterpret_cast<const vec_t*>(base + 2 * D), vc);
  float dlc;
  {
    float oc[DPL];
    V::unpack(*reinterpret_cast<const vec_t*>(obase), oc);
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) t = fmaf(doc[i], oc[i], t);
    dlc = group_sum<DPL>(t);
  }
  const float Lc2 = lse[((size_t)b * H + h) * T] * kLog2e;            // cls-row lse in log2 units

  float dqc[DPL], dkc[DPL], dvc[DPL], dqs[RIDER ? DPL : 1];
#pragma unroll
  for (int i = 0; i < DPL; ++i) { dqc[i] = 0.f; dkc[i] = 0.f; dvc[i] = 0.f; }
  if (RIDER) {
#pragma unroll
    for (int i = 0; i < DPL; ++i) dqs[i] = 0.f;
  }
  if (chunk == 0 && n_sub == 0) {     // the cls key inside the CLS row, once per (b,h)
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) { s = fmaf(qc[i], kc[i], s); dp = fmaf(doc[i], vc[i], dp); }
    s = group_sum<DPL>(s) * (0.125f * kLog2e);
    dp = group_sum<DPL>(dp);
    const float p = __builtin_amdgcn_exp2f(s - Lc2), ds = p * (dp - dlc) * 0.125f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) { dvc[i] = p * doc[i]; dkc[i] = ds * qc[i]; dqc[i] = ds * kc[i]; }
  }

  const int n_end = min(N, (chunk + 1) * NCH);
#pragma unroll 1
  for (int n = chunk * NCH + n_sub; n < n_end; n += NPB) {
    vec_t kk[F], vv[F], qq[F], gg[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const int tok = 1 + f * N + n;
      const E* p = base + (size_t)tok * ts;
      qq[f] = *reinterpret_cast<const vec_t*>(p);
      kk[f] = *reinterpret_cast<const vec_t*>(p + D);
      vv[f] = *reinterpret_cast<const vec_t*>(p + 2 * D);
      gg[f] = *reinterpret_cast<const vec_t*>(dobase + (size_t)tok * D);
    }
    float dk[F][DPL], dv[F][DPL];
    // CLS-row terms for this location's F keys
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float kf[DPL], vf[DPL];
      V::unpack(kk[f], kf);
      V::unpack(vv[f], vf);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int i = 0; i < DPL; ++i) { s = fmaf(qc[i], kf[i], s); dp = fmaf(doc[i], vf[i], dp); }
      s = group_sum<DPL>(s) * (0.125f * kLog2e);
      dp = group_sum<DPL>(dp);
      const float p = __builtin_amdgcn_exp2f(s - Lc2), ds = p * (dp - dlc) * 0.125f;
#pragma unroll
      for (int i = 0; i < DPL; ++i) {
        dv[f][i] = p * doc[i];
        dk[f][i] = ds * qc[i];
        dqc[i] = fmaf(ds, kf[i], dqc[i]);
      }
    }
    // the F patch queries
#pragma unroll
    for (int fq = 0; fq < F; ++fq) {
      const int tok = 1 + fq * N + n;
      float q[DPL], go[DPL];
      V::unpack(qq[fq], q);
      V::unpack(gg[fq], go);
      float s[F + 1], dp[F + 1];
      {
        float a = 0.f, d = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; ++i) { a = fmaf(q[i], kc[i], a); d = fmaf(go[i], vc[i], d); }
        s[0] = group_sum<DPL>(a) * (0.125f * kLog2e);
        dp[0] = group_sum<DPL>(d);
      }
      float mx = s[0];
#pragma unroll
      for (int f = 0; f < F; ++f) {
        s[f + 1] = group_sum<DPL>(dotp<E, DPL>(q, kk[f])) * (0.125f * kLog2e);
        dp[f + 1] = group_sum<DPL>(dotp<E, DPL>(go, vv[f]));
        mx = fmaxf(mx, s[f + 1]);
      }
      float l = 0.f;
#pragma unroll
      for (int j = 0; j <= F; ++j) { s[j] = __builtin_amdgcn_exp2f(s[j] - mx); l += s[j]; }
      const float linv = __builtin_amdgcn_rcpf(l);
      // delta_q = dO_q . O_q = sum_j P_qj dP_qj: all F+1 keys are in registers, the O rows are never read
      float dlt = 0.f;
#pragma unroll
      for (int j = 0; j <= F; ++j) dlt = fmaf(s[j], dp[j], dlt);
      dlt *= linv;
      float dq[DPL];
      {
        const float p = s[0] * linv, ds = p * (dp[0] - dlt) * 0.125f;
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
          dq[i] = ds * kc[i];
          dkc[i] = fmaf(ds, q[i], dkc[i]);
          dvc[i] = fmaf(p, go[i], dvc[i]);
        }
      }
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float p = s[f + 1] * linv, ds = p * (dp[f + 1] - dlt) * 0.125f;
        float kf[DPL];
        V::unpack(kk[f], kf);
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
          dq[i] = fmaf(ds, kf[i], dq[i]);
          dk[f][i] = fmaf(ds, q[i], dk[f][i]);
          dv[f][i] = fmaf(p, go[i], dv[f][i]);
        }
      }
      *reinterpret_cast<vec_t*>(gbase + (size_t)tok * ts) = V::pack(dq);
      if (RIDER) {
#pragma unroll
        for (int i = 0; i < DPL; ++i) dqs[i] += dq[i];
      }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
      E* p = gbase + (size_t)(1 + f * N + n) * ts;
      *reinterpret_cast<vec_t*>(p + D) = V::pack(dk[f]);
      *reinterpret_cast<vec_t*>(p + 2 * D) = V::pack(dv[f]);
    }
  }

  constexpr int RS = (RIDER ? 4 : 3) * DPL;
  float* mine = smem + ((size_t)(n_sub * H + h) * LPP + dl) * RS;
#pragma unroll
  for (int i = 0; i < DPL; ++i) {
    mine[i] = dqc[i]; mine[DPL + i] = dkc[i]; mine[2 * DPL + i] = dvc[i];
    if (RIDER) mine[3 * DPL + i] = dqs[i] + dqc[i];      // patch rows + this thread's share of the cls query's dq
  }
  __syncthreads();
  if (n_sub == 0) {
    float acc[RS];
#pragma unroll
    for (int i = 0; i < RS; ++i) acc[i] = 0.f;
    for (int s = 0; s < NPB; ++s) {
      const float* r = smem + ((size_t)(s * H + h) * LPP + dl) * RS;
#pragma unroll
      for (int i = 0; i < RS; ++i) acc[i] += r[i];
    }
    // this chunk's record of the cls token's d(q | k | v): slot `chunk` of the (b, h) partial slab, one writer, plain
    // stores; cls_grad_finalize_kernel adds the NC slots up in order (round 6: deterministic, no f32 atomics)
    float* dst = atom_ws + (((size_t)b * H + h) * NC + chunk) * 192 + dl * DPL;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      dst[i] = acc[i];
      dst[64 + i] = acc[DPL + i];
      dst[128 + i] = acc[2 * DPL + i];
    }
    if (RIDER) {
      float* qd = dq_part + (size_t)blockIdx.x * D + h * 64 + dl * DPL;
#pragma unroll
      for (int i = 0; i < DPL; ++i) qd[i] = acc[3 * DPL + i];
    }
  }
}

int gcd_int(int a, int b) { return b ? gcd_int(b, a % b) : a; }

// channels per lane: 4 (16 lanes per problem) up to 4 frames, 2 (32 lanes per problem) from 8 frames on, where the
// F key / value / query / dO rows would not fit the register file at 4 channels per lane
inline int time_dpl(int F) { return F >= 8 ? 2 : 4; }

struct TimeGeom { int NPB, NCH, NC, block; bool ok; };

TimeGeom time_geometry(int N, int H, int dpl) {
  TimeGeom g{};
  const int per_n = (64 / dpl) * H;
  g.NPB = 64 / gcd_int(per_n, 64);
  if (per_n * g.NPB < 128) g.NPB *= 2;
  g.block = per_n * g.NPB;
  g.ok = g.block <= (dpl == 2 ? 512 : 256);
  int nch = g.NPB * 8;                       // >= 8 locations per thread group: amortise the cls/LDS epilogue
  while ((N + nch - 1) / nch > 64) nch *= 2;  // at most 64 partial records per (b,h)
  g.NCH = nch;
  g.NC = (N + nch - 1) / nch;
  return g;
}

}  // namespace

void lvl_launch_cls_combine(const float* ws, void* out, float* lse, int B, int H, int nparts, int T, int dtype,
                            hipStream_t st);
void lvl_launch_cls_grad_finalize(const float* atom_ws, void* dqkv, int B, int T, int H, int nslots, int dtype,
                                  hipStream_t st);

bool lvl_time_fast_supported(int F, int N, int H) {
  if (!(F == 1 || F == 2 || F == 3 || F == 4 || F == 8 || F == 16)) return false;
  return time_geometry(N, H, time_dpl(F)).ok;
}

int lvl_time_fast_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, int dtype,
                      hipStream_t st) {
  const int dpl = time_dpl(F);
  const TimeGeom g = time_geometry(N, H, dpl);
  if (!g.ok) return lvl_fail(LVL_ENOSYS, "time_fast_fwd: unsupported head count %d", H);
  const size_t shmem = (size_t)g.NPB * H * (64 / dpl) * (2 + dpl) * sizeof(float);
  const dim3 grid((unsigned)(B * g.NC)), block(g.block);
#define TIME_FWD(FF, DD)                                                                                          \
  do {                                                                                                            \
    if (dtype == LVL_F32)                                                                                         \
      hipLaunchKernelGGL((time_fwd_kernel<float, FF, DD>), grid, block, shmem, st, (const float*)qkv, (float*)out, \
                         lse, ws, N, H, g.NPB, g.NCH, g.NC);                                                      \
    else                                                                                                          \
      hipLaunchKernelGGL((time_fwd_kernel<uint16_t, FF, DD>), grid, block, shmem, st, (const uint16_t*)qkv,        \
                         (uint16_t*)out, lse, ws, N, H, g.NPB, g.NCH, g.NC);                                      \
  } while (0)
  switch (F) {
    case 1: TIME_FWD(1, 4); break;
    case 2: TIME_FWD(2, 4); break;
    case 3: TIME_FWD(3, 4); break;
    case 4: TIME_FWD(4, 4); break;
    case 8: TIME_FWD(8, 2); break;
    case 16: TIME_FWD(16, 2); break;
    default: return lvl_fail(LVL_ENOSYS, "time_fast_fwd: unsupported frame count %d", F);
  }
#undef TIME_FWD
  LVL_CHECK_LAUNCH("time_fwd");
  lvl_launch_cls_combine(ws, out, lse, B, H, g.NC, 1 + F * N, dtype, st);
  LVL_CHECK_LAUNCH("cls_combine");
  return LVL_OK;
}

// -1 (default): measured choice per shape -- 2 at F = 4 (the benched TSF-B shape: 0.485 ms per backward + bias gradient
// against 0.59-0.62 without the rider and 0.61 with it at 3 waves per SIMD; the spill-free kernel is faster than the
// 13-spill one even before the saved pass, profiles/r05_time_bwd_rider.txt), 1 elsewhere (those instantiations have
// registers to spare); 0: no rider (the q third of d(qkv bias) is reduced from dqkv afterwards); 1 / 2: see time_bwd_kernel
static std::atomic<int> g_time_rider{-2};          // -2: not read yet (LAVILA_TIME_BWD_RIDER in the environment, A/B runs)
static int time_rider_mode(int F) {
  int m = g_time_rider.load(std::memory_order_relaxed);
  if (m == -2) {
    const char* e = getenv("LAVILA_TIME_BWD_RIDER");
    m = (e && e[0] >= '0' && e[0] <= '2' && !e[1]) ? e[0] - '0' : -1;
    g_time_rider.store(m, std::memory_order_relaxed);
  }
  return m >= 0 ? m : (F == 4 ? 2 : 1);
}

extern "C" int lvl_debug_time_bwd_rider(int mode) {
  if (mode < -1 || mode > 2) return lvl_fail(LVL_EINVAL, "debug_time_bwd_rider: mode must be -1 (auto), 0, 1 or 2");
  g_time_rider.store(mode, std::memory_order_relaxed);
  return LVL_OK;
}

bool lvl_time_fast_bwd_supported(int F, int N, int H) {
  if (!(F == 1 || F == 2 || F == 3 || F == 4 || F == 8 || F == 16)) return false;
  return time_geometry(N, H, time_dpl(F)).ok;
}

// rows of the dq column-sum slab (one per workgroup); 0 = this family runs without the rider
int lvl_time_fast_bwd_dq_part_rows(int B, int F, int N, int H) {
  const TimeGeom g = time_geometry(N, H, time_dpl(F));
  return (g.ok && time_rider_mode(F) != 0) ? B * g.NC : 0;
}

// ws layout: delta [B*H*T] f32 (unused here), then the cls token's partial records [B*H][NC][192] f32
// dq_part (nullable): [lvl_time_fast_bwd_dq_part_rows, H*64] f32 partial column sums of dq (bias-gradient rider)
int lvl_time_fast_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      float* dq_part, int B, int F, int N, int H, int dtype, hipStream_t st) {
  const int dpl = time_dpl(F);
  const TimeGeom g = time_geometry(N, H, dpl);
  if (!g.ok) return lvl_fail(LVL_ENOSYS, "time_fast_bwd: unsupported head count %d", H);
  const int T = 1 + F * N;
  float* atom_ws = ws + (size_t)B * H * T;            // every (b, h, chunk) slot is written whole by its workgroup: no zeroing
  const int rider = dq_part ? time_rider_mode(F) : 0;
  const size_t shmem = (size_t)g.NPB * H * (64 / dpl) * (rider ? 4 : 3) * dpl * sizeof(float);
  const dim3 grid((unsigned)(B * g.NC)), block(g.block);
#define TIME_BWD_R(FF, DD, RR)                                                                                            \
  do {                                                                                                              \
    if (dtype == LVL_F32)                                                                                           \
      hipLaunchKernelGGL((time_bwd_kernel<float, FF, DD, RR>), grid, block, shmem, st, (const float*)qkv,            \
                         (const float*)out, (const float*)dout, lse, (float*)dqkv, atom_ws, dq_part, N, H, g.NPB,      \
                         g.NCH, g.NC);                                                                              \
    else                                                                                                            \
      hipLaunchKernelGGL((time_bwd_kernel<uint16_t, FF, DD, RR>), grid, block, shmem, st, (const uint16_t*)qkv,      \
                         (const uint16_t*)out, (const uint16_t*)dout, lse, (uint16_t*)dqkv, atom_ws, dq_part, N, H,  \
                         g.NPB, g.NCH, g.NC);                                                                       \
  } while (0)
#define TIME_BWD(FF, DD)                         \
  do {                                           \
    if (rider == 2) TIME_BWD_R(FF, DD, 2);       \
    else if (rider == 1) TIME_BWD_R(FF, DD, 1);  \
    else TIME_BWD_R(FF, DD, 0);                  \
  } while (0)
  switch (F) {
    case 1: TIME_BWD(1, 4); break;
    case 2: TIME_BWD(2, 4); break;
    case 3: TIME_BWD(3, 4); break;
    case 4: TIME_BWD(4, 4); break;
    case 8: TIME_BWD(8, 2); break;
    case 16: TIME_BWD(16, 2); break;
    default: return lvl_fail(LVL_ENOSYS, "time_fast_bwd: unsupported frame count %d", F);
  }
#undef TIME_BWD
#undef TIME_BWD_R
  LVL_CHECK_LAUNCH("time_bwd");
  lvl_launch_cls_grad_finalize(atom_ws, dqkv, B, T, H, g.NC, dtype, st);
  LVL_CHECK_LAUNCH("cls_grad_finalize");
  return LVL_OK;
}
