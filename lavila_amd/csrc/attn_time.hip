// Time-mode divided attention forward (bf16 in/out, f32 arithmetic), gfx950.
//
// Per (sample b, location n, head h): F queries x (1 cls + F) keys, head dim 64 (timesformer.py:121-131
// with the '(b n) f d' grouping :302-303). 2.5 flop/B at F=4: purely HBM-bound, nothing for the matrix
// cores to do. Eight lanes share one problem (8 channels each, one 16-B vector per row), so a wave's load of
// "row f of 8 adjacent heads" is one contiguous 1-KB segment; K and V of the group stay packed in registers;
// every row of qkv is read once and every row of out is written once.
// The CLS query (token 0) attends to ALL keys: each thread group folds its own F keys into a running
// flash-style partial (max, sum, acc) for its head while the rows are in registers; partials are merged per
// workgroup through LDS and across workgroups by cls_combine_kernel (attn_space_mfma.hip).
#include "common.h"

namespace {

constexpr int CLS_REC = 66;

__device__ __forceinline__ void unpack8(const uint4& a, float (&v)[8]) {
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}

__device__ __forceinline__ float dot8(const float (&a)[8], const uint4& b) {
  float v[8];
  unpack8(b, v);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s = fmaf(a[i], v[i], s);
  return s;
}

__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

// block = 8 * H * NPB threads; thread group (8 lanes) = fixed head h, location slot n_sub
template <int F>
__global__ __launch_bounds__(256) void time_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                float* __restrict__ lse, float* __restrict__ cls_ws, int N, int H, int NPB, int NCH,
                                int NC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [NPB][H][8 lanes][10]
  const int tid = threadIdx.x, dl = tid & 7, grp = tid >> 3;
  const int h = grp % H, n_sub = grp / H;
  const int chunk = blockIdx.x % NC, b = blockIdx.x / NC;
  const int D = H * 64, T = 1 + F * N;
  const size_t ts = (size_t)3 * D;
  const uint16_t* base = qkv + (size_t)b * T * ts + h * 64 + dl * 8;
  uint16_t* obase = out + (size_t)b * T * D + h * 64 + dl * 8;
  float* lrow = lse + ((size_t)b * H + h) * T;

  const uint4 kc = *reinterpret_cast<const uint4*>(base + D);           // cls key / value / query of this head
  const uint4 vc = *reinterpret_cast<const uint4*>(base + 2 * D);
  float qc[8];
  unpack8(*reinterpret_cast<const uint4*>(base), qc);
#pragma unroll
  for (int i = 0; i < 8; ++i) qc[i] *= 0.125f;

  float cm = -INFINITY, cl = 0.f, ca[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (chunk == 0 && n_sub == 0) {           // the cls key itself enters the CLS row exactly once per (b,h)
    cm = group8_sum(dot8(qc, kc));
    cl = 1.f;
    unpack8(vc, ca);
  }

  const int n_end = min(N, (chunk + 1) * NCH);
#pragma unroll 1
  for (int n = chunk * NCH + n_sub; n < n_end; n += NPB) {
    uint4 kk[F], vv[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const uint16_t* p = base + (size_t)(1 + f * N + n) * ts;
      kk[f] = *reinterpret_cast<const uint4*>(p + D);
      vv[f] = *reinterpret_cast<const uint4*>(p + 2 * D);
    }
    // CLS-query partial over this location's F keys
    {
      float s[F], mx = cm;
#pragma unroll
      for (int f = 0; f < F; ++f) { s[f] = group8_sum(dot8(qc, kk[f])); mx = fmaxf(mx, s[f]); }
      const float al = __expf(cm - mx);
      cl *= al;
#pragma unroll
      for (int i = 0; i < 8; ++i) ca[i] *= al;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float p = __expf(s[f] - mx);
        float v[8];
        unpack8(vv[f], v);
        cl += p;
#pragma unroll
        for (int i = 0; i < 8; ++i) ca[i] = fmaf(p, v[i], ca[i]);
      }
      cm = mx;
    }
    // the F patch queries of this location
#pragma unroll
    for (int fq = 0; fq < F; ++fq) {
      float q[8];
      unpack8(*reinterpret_cast<const uint4*>(base + (size_t)(1 + fq * N + n) * ts), q);
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] *= 0.125f;
      float s[F + 1];
      s[0] = group8_sum(dot8(q, kc));
      float mx = s[0];
#pragma unroll
      for (int f = 0; f < F; ++f) { s[f + 1] = group8_sum(dot8(q, kk[f])); mx = fmaxf(mx, s[f + 1]); }
      float o[8], v[8];
      float p = __expf(s[0] - mx), l = p;
      unpack8(vc, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = p * v[i];
#pragma unroll
      for (int f = 0; f < F; ++f) {
        p = __expf(s[f + 1] - mx);
        l += p;
        unpack8(vv[f], v);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaf(p, v[i], o[i]);
      }
      const float linv = 1.0f / l;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] *= linv;
      const int tok = 1 + fq * N + n;
      Elem<bf16_t>::store8(reinterpret_cast<bf16_t*>(obase + (size_t)tok * D), o);
      if (dl == 0) lrow[tok] = mx + __logf(l);
    }
  }

  // merge the NPB location slots of each head, one record per (b, h, chunk)
  float* mine = smem + ((size_t)(n_sub * H + h) * 8 + dl) * 10;
  mine[0] = cm; mine[1] = cl;
#pragma unroll
  for (int i = 0; i < 8; ++i) mine[2 + i] = ca[i];
  __syncthreads();
  if (n_sub == 0) {
    float M = -INFINITY;
    for (int s = 0; s < NPB; ++s) M = fmaxf(M, smem[((size_t)(s * H + h) * 8 + dl) * 10]);
    float Ls = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < NPB; ++s) {
      const float* r = smem + ((size_t)(s * H + h) * 8 + dl) * 10;
      const float w = r[0] == -INFINITY ? 0.f : __expf(r[0] - M);
      Ls = fmaf(r[1], w, Ls);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(r[2 + i], w, acc[i]);
    }
    float* rec = cls_ws + (((size_t)b * H + h) * NC + chunk) * CLS_REC;
    if (dl == 0) { rec[0] = M; rec[1] = Ls; }
#pragma unroll
    for (int i = 0; i < 8; ++i) rec[2 + dl * 8 + i] = acc[i];
  }
}


// ---- backward ------------------------------------------------------------------------------------------------
// Same thread geometry as the forward. Everything of a (b, n, h) problem is thread-group local: the softmax is
// recomputed from the F+1 keys in registers (no saved statistics needed), dq/dk/dv rows of the patch tokens are
// written exactly once. Gradient of the cls key/value (shared by all N locations) and of the cls query (which
// attends to every key; its rank-1 terms are folded into dk/dv here) is accumulated per thread group, merged
// per workgroup through LDS and added to an f32 workspace with atomics (finalised by cls_grad_finalize_kernel).
template <int F>
__global__ __launch_bounds__(256) void time_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                       const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                       uint16_t* __restrict__ dqkv, float* __restrict__ atom_ws, int N,
                                                       int H, int NPB, int NCH, int NC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [NPB][H][8 lanes][24]
  const int tid = threadIdx.x, dl = tid & 7, grp = tid >> 3;
  const int h = grp % H, n_sub = grp / H;
  const int chunk = blockIdx.x % NC, b = blockIdx.x / NC;
  const int D = H * 64, T = 1 + F * N;
  const size_t ts = (size_t)3 * D;
  const uint16_t* base = qkv + (size_t)b * T * ts + h * 64 + dl * 8;
  uint16_t* gbase = dqkv + (size_t)b * T * ts + h * 64 + dl * 8;
  const uint16_t* obase = out + (size_t)b * T * D + h * 64 + dl * 8;
  const uint16_t* dobase = dout + (size_t)b * T * D + h * 64 + dl * 8;

  const uint4 kcp = *reinterpret_cast<const uint4*>(base + D);
  const uint4 vcp = *reinterpret_cast<const uint4*>(base + 2 * D);
  float qc[8], doc[8], kc[8], vc[8];
  unpack8(*reinterpret_cast<const uint4*>(base), qc);              // raw cls query
  unpack8(*reinterpret_cast<const uint4*>(dobase), doc);           // d out of the cls row
  unpack8(kcp, kc);
  unpack8(vcp, vc);
  float dlc;
  {
    float oc[8];
    unpack8(*reinterpret_cast<const uint4*>(obase), oc);
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t = fmaf(doc[i], oc[i], t);
    dlc = group8_sum(t);
  }
  const float Lc = lse[((size_t)b * H + h) * T];

  float dqc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dkc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dvc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (chunk == 0 && n_sub == 0) {     // the cls key inside the CLS row, once per (b,h)
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s = fmaf(qc[i], kc[i], s); dp = fmaf(doc[i], vc[i], dp); }
    s = group8_sum(s) * 0.125f;
    dp = group8_sum(dp);
    const float p = __expf(s - Lc), ds = p * (dp - dlc) * 0.125f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { dvc[i] = p * doc[i]; dkc[i] = ds * qc[i]; dqc[i] = ds * kc[i]; }
  }

  const int n_end = min(N, (chunk + 1) * NCH);
#pragma unroll 1
  for (int n = chunk * NCH + n_sub; n < n_end; n += NPB) {
    uint4 kk[F], vv[F];
    float dk[F][8], dv[F][8];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const uint16_t* p = base + (size_t)(1 + f * N + n) * ts;
      kk[f] = *reinterpret_cast<const uint4*>(p + D);
      vv[f] = *reinterpret_cast<const uint4*>(p + 2 * D);
    }
    // CLS-row terms for this location's F keys
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float kf[8], vf[8];
      unpack8(kk[f], kf);
      unpack8(vv[f], vf);
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { s = fmaf(qc[i], kf[i], s); dp = fmaf(doc[i], vf[i], dp); }
      s = group8_sum(s) * 0.125f;
      dp = group8_sum(dp);
      const float p = __expf(s - Lc), ds = p * (dp - dlc) * 0.125f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        dv[f][i] = p * doc[i];
        dk[f][i] = ds * qc[i];
        dqc[i] = fmaf(ds, kf[i], dqc[i]);
      }
    }
    // the F patch queries
#pragma unroll
    for (int fq = 0; fq < F; ++fq) {
      const int tok = 1 + fq * N + n;
      float q[8], go[8], oo[8];
      unpack8(*reinterpret_cast<const uint4*>(base + (size_t)tok * ts), q);
      unpack8(*reinterpret_cast<const uint4*>(dobase + (size_t)tok * D), go);
      unpack8(*reinterpret_cast<const uint4*>(obase + (size_t)tok * D), oo);
      float dlt = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) dlt = fmaf(go[i], oo[i], dlt);
      dlt = group8_sum(dlt);
      float s[F + 1], dp[F + 1];
      {
        float a = 0.f, d = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a = fmaf(q[i], kc[i], a); d = fmaf(go[i], vc[i], d); }
        s[0] = group8_sum(a) * 0.125f;
        dp[0] = group8_sum(d);
      }
      float mx = s[0];
#pragma unroll
      for (int f = 0; f < F; ++f) {
        s[f + 1] = group8_sum(dot8(q, kk[f])) * 0.125f;
        dp[f + 1] = group8_sum(dot8(go, vv[f]));
        mx = fmaxf(mx, s[f + 1]);
      }
      float l = 0.f;
#pragma unroll
      for (int j = 0; j <= F; ++j) { s[j] = __expf(s[j] - mx); l += s[j]; }
      const float linv = 1.0f / l;
      float dq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      {
        const float p = s[0] * linv, ds = p * (dp[0] - dlt) * 0.125f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          dq[i] = ds * kc[i];
          dkc[i] = fmaf(ds, q[i], dkc[i]);
          dvc[i] = fmaf(p, go[i], dvc[i]);
        }
      }
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float p = s[f + 1] * linv, ds = p * (dp[f + 1] - dlt) * 0.125f;
        float kf[8];
        unpack8(kk[f], kf);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          dq[i] = fmaf(ds, kf[i], dq[i]);
          dk[f][i] = fmaf(ds, q[i], dk[f][i]);
          dv[f][i] = fmaf(p, go[i], dv[f][i]);
        }
      }
      Elem<bf16_t>::store8(reinterpret_cast<bf16_t*>(gbase + (size_t)tok * ts), dq);
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
      uint16_t* p = gbase + (size_t)(1 + f * N + n) * ts;
      Elem<bf16_t>::store8(reinterpret_cast<bf16_t*>(p + D), dk[f]);
      Elem<bf16_t>::store8(reinterpret_cast<bf16_t*>(p + 2 * D), dv[f]);
    }
  }

  float* mine = smem + ((size_t)(n_sub * H + h) * 8 + dl) * 24;
#pragma unroll
  for (int i = 0; i < 8; ++i) { mine[i] = dqc[i]; mine[8 + i] = dkc[i]; mine[16 + i] = dvc[i]; }
  __syncthreads();
  if (n_sub == 0) {
    float acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = 0.f;
    for (int s = 0; s < NPB; ++s) {
      const float* r = smem + ((size_t)(s * H + h) * 8 + dl) * 24;
#pragma unroll
      for (int i = 0; i < 24; ++i) acc[i] += r[i];
    }
    float* dst = atom_ws + ((size_t)b * H + h) * 192 + dl * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      atomicAdd(dst + i, acc[i]);
      atomicAdd(dst + 64 + i, acc[8 + i]);
      atomicAdd(dst + 128 + i, acc[16 + i]);
    }
  }
}

int gcd_int(int a, int b) { return b ? gcd_int(b, a % b) : a; }

struct TimeGeom { int NPB, NCH, NC, block; bool ok; };

TimeGeom time_geometry(int N, int H) {
  TimeGeom g{};
  const int per_n = 8 * H;
  g.NPB = 64 / gcd_int(per_n, 64);
  if (per_n * g.NPB < 128) g.NPB *= 2;
  g.block = per_n * g.NPB;
  g.ok = g.block <= 256;
  int nch = g.NPB * 8;                       // >= 8 locations per thread group: amortise the cls/LDS epilogue
  while ((N + nch - 1) / nch > 64) nch *= 2;  // at most 64 partial records per (b,h)
  g.NCH = nch;
  g.NC = (N + nch - 1) / nch;
  return g;
}

}  // namespace

void lvl_launch_cls_combine(const float* ws, void* out, float* lse, int B, int H, int nparts, int T, hipStream_t st);

bool lvl_time_fast_supported(int F, int N, int H) {
  if (!(F == 1 || F == 2 || F == 3 || F == 4 || F == 8 || F == 16)) return false;
  return time_geometry(N, H).ok;
}

int lvl_time_fast_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st) {
  const TimeGeom g = time_geometry(N, H);
  if (!g.ok) return lvl_fail(LVL_ENOSYS, "time_fast_fwd: unsupported head count %d", H);
  const size_t shmem = (size_t)g.NPB * H * 8 * 10 * sizeof(float);
  const dim3 grid((unsigned)(B * g.NC)), block(g.block);
#define TIME_FWD(FF)                                                                                            \
  hipLaunchKernelGGL((time_fwd_kernel<FF>), grid, block, shmem, st, (const uint16_t*)qkv, (uint16_t*)out, lse, ws, N, H, \
                     g.NPB, g.NCH, g.NC)
  switch (F) {
    case 1: TIME_FWD(1); break;
    case 2: TIME_FWD(2); break;
    case 3: TIME_FWD(3); break;
    case 4: TIME_FWD(4); break;
    case 8: TIME_FWD(8); break;
    case 16: TIME_FWD(16); break;
    default: return lvl_fail(LVL_ENOSYS, "time_fast_fwd: unsupported frame count %d", F);
  }
#undef TIME_FWD
  LVL_CHECK_LAUNCH("time_fwd");
  lvl_launch_cls_combine(ws, out, lse, B, H, g.NC, 1 + F * N, st);
  LVL_CHECK_LAUNCH("cls_combine");
  return LVL_OK;
}

void lvl_launch_cls_grad_finalize(const float* atom_ws, void* dqkv, int B, int T, int H, hipStream_t st);

bool lvl_time_fast_bwd_supported(int F, int N, int H) {
  if (!(F == 1 || F == 2 || F == 3 || F == 4 || F == 8)) return false;
  return time_geometry(N, H).ok;
}

// ws layout: delta [B*H*T] f32 (unused here), then atomics [B*H*192] f32
int lvl_time_fast_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int F, int N, int H, hipStream_t st) {
  const TimeGeom g = time_geometry(N, H);
  if (!g.ok) return lvl_fail(LVL_ENOSYS, "time_fast_bwd: unsupported head count %d", H);
  const int T = 1 + F * N;
  float* atom_ws = ws + (size_t)B * H * T;
  hipError_t e = hipMemsetAsync(atom_ws, 0, (size_t)B * H * 192 * sizeof(float), st);
  if (e != hipSuccess) return lvl_fail(LVL_EHIP, "time_bwd memset: %s", hipGetErrorString(e));
  const size_t shmem = (size_t)g.NPB * H * 8 * 24 * sizeof(float);
  const dim3 grid((unsigned)(B * g.NC)), block(g.block);
#define TIME_BWD(FF)                                                                                              \
  hipLaunchKernelGGL((time_bwd_kernel<FF>), grid, block, shmem, st, (const uint16_t*)qkv, (const uint16_t*)out,   \
                     (const uint16_t*)dout, lse, (uint16_t*)dqkv, atom_ws, N, H, g.NPB, g.NCH, g.NC)
  switch (F) {
    case 1: TIME_BWD(1); break;
    case 2: TIME_BWD(2); break;
    case 3: TIME_BWD(3); break;
    case 4: TIME_BWD(4); break;
    case 8: TIME_BWD(8); break;
    default: return lvl_fail(LVL_ENOSYS, "time_fast_bwd: unsupported frame count %d", F);
  }
#undef TIME_BWD
  LVL_CHECK_LAUNCH("time_bwd");
  lvl_launch_cls_grad_finalize(atom_ws, dqkv, B, T, H, st);
  LVL_CHECK_LAUNCH("cls_grad_finalize");
  return LVL_OK;
}
