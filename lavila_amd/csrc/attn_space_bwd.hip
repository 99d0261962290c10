// Space-mode divided attention backward on the matrix cores (bf16, f32 accumulate), gfx950.
//
// With every key of a (sample, frame, head) group LDS-resident the backward recomputes P = exp(S - lse)
// from the saved row log-sum-exp (no max/sum pass) and needs four contractions:
//     dP = dO V^T (over d)     dS = P o (dP - delta),  delta_q = dO_q . O_q
//     dQ = dS K   (over keys)  dK = dS^T Q (over queries)   dV = P^T dO (over queries)
// A 16x16x32 MFMA wants its contraction index contiguous per lane, so contractions over keys / queries
// read TRANSPOSED LDS images (Kt[d][key], Qt[d][q], dOt[d][q]) as B operands while the A operand is the
// freshly computed tile itself: the C layout of S^T (resp. S) puts one query (resp. key) per lane, which
// is exactly the A-fragment layout with a permuted k-order (same trick as the forward's P.V).
// Two kernels, each one workgroup of 8 waves per (b, f, h):
//   dq kernel : waves own 16-query tiles, all keys resident (Ks, Vs row-major, Kt transposed) -> dQ, delta
//   dkv kernel: waves own 16-key tiles, all queries resident (Qs, dOs row-major, Qt, dOt transposed)
//               -> dK, dV; also folds in the CLS query's rank-1 contributions to dK/dV (it attends to every
//               key, timesformer.py:116-119) and accumulates d(cls q) and d(cls k,v) -- which receive
//               gradient from every frame -- with f32 atomics into a workspace finalised by a tiny kernel.
#include "attn_mfma_common.h"

namespace {

using namespace attn_mfma;

// ------------------------------------------------------------------------------------------------------------
// dQ kernel
// ------------------------------------------------------------------------------------------------------------
template <int NKT> struct DqLds {
  static constexpr int KROWS = NKT * 16, LDK = NKT * 16 + 8;
  static constexpr int ks_off = 0;
  static constexpr int vs_off = ks_off + KROWS * KS * 2;
  static constexpr int kt_off = vs_off + KROWS * KS * 2;
  static constexpr int ot_off = kt_off + 64 * LDK * 2;
  static constexpr int total = ot_off + 8 * 16 * OS * 2;
};

template <int NKT, bool TEXT>
__global__ __launch_bounds__(512) void space_bwd_dq_kernel(const uint16_t* __restrict__ qkv,
                                                           const uint16_t* __restrict__ out,
                                                           const uint16_t* __restrict__ dout,
                                                           const float* __restrict__ lse, uint16_t* __restrict__ dqkv,
                                                           float* __restrict__ delta, int F, int N, int H) {
  using L = DqLds<NKT>;
  constexpr int LDK = L::LDK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* Ks = reinterpret_cast<uint16_t*>(smem + L::ks_off);
  uint16_t* Vs = reinterpret_cast<uint16_t*>(smem + L::vs_off);
  uint16_t* Kt = reinterpret_cast<uint16_t*>(smem + L::kt_off);
  uint16_t* Ot = reinterpret_cast<uint16_t*>(smem + L::ot_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % H, f = (blockIdx.x / H) % F, b = blockIdx.x / (H * F);
  const int D = H * 64, T = TEXT ? N : 1 + F * N, nkeys = TEXT ? N : N + 1, tok0 = TEXT ? 0 : 1 + f * N;
  const size_t ts = (size_t)3 * D;
  const uint16_t* base = qkv + (size_t)b * T * ts + h * 64;
  const uint16_t* obase = out + (size_t)b * T * D + h * 64;
  const uint16_t* dobase = dout + (size_t)b * T * D + h * 64;

  // fragments of this wave's first query tile: issued before the staging so that both are in flight together
  const int c = lane & 15, g = lane >> 4;
  auto tok_of = [&](int qt) { const int qr = qt * 16 + c; return tok0 + (qr < N ? qr : N - 1); };
  uint4 nq0, nq1, ng0, ng1, ny0, ny1;
  auto load_frags = [&](int qt) {
    const int tk = tok_of(qt);
    const uint16_t* qp = base + (size_t)tk * ts + g * 8;
    nq0 = *reinterpret_cast<const uint4*>(qp); nq1 = *reinterpret_cast<const uint4*>(qp + 32);
    ng0 = *reinterpret_cast<const uint4*>(dobase + (size_t)tk * D + g * 8);
    ng1 = *reinterpret_cast<const uint4*>(dobase + (size_t)tk * D + g * 8 + 32);
    ny0 = *reinterpret_cast<const uint4*>(obase + (size_t)tk * D + g * 8);
    ny1 = *reinterpret_cast<const uint4*>(obase + (size_t)tk * D + g * 8 + 32);
  };
  load_frags(wave * 16 < N ? wave : 0);

  stage_rows2<512, (L::KROWS + 63) / 64>(
      Ks, Kt, [&](int r) { return base + (size_t)(TEXT ? r : (r == 0 ? 0 : tok0 + r - 1)) * ts + D; },
      Vs, nullptr, [&](int r) { return base + (size_t)(TEXT ? r : (r == 0 ? 0 : tok0 + r - 1)) * ts + 2 * D; },
      LDK, L::KROWS, nkeys, tid);
  __syncthreads();

  uint16_t* ot = Ot + wave * 16 * OS;
#pragma unroll 1
  for (int qt = wave; qt * 16 < N; qt += 8) {
    const int qrow = qt * 16 + c;
    const int tok = tok_of(qt);
    const uint4 q0 = nq0, q1 = nq1, g0 = ng0, g1 = ng1, y0 = ny0, y1 = ny1;
    if ((qt + 8) * 16 < N) load_frags(qt + 8);
    float dl;
    {
      float a[8], bb[8];
      Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&g0), a);
      Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&y0), bb);
      dl = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) dl = fmaf(a[i], bb[i], dl);
      Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&g1), a);
      Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&y1), bb);
#pragma unroll
      for (int i = 0; i < 8; ++i) dl = fmaf(a[i], bb[i], dl);
      dl += __shfl_xor(dl, 16, 64);
      dl += __shfl_xor(dl, 32, 64);
    }
    const size_t srow = ((size_t)b * H + h) * T + tok;
    const float Lq = lse[srow];
    if (g == 0 && qrow < N) delta[srow] = dl;

    // sweeps of independent MFMAs over the key tiles (see attn_space_mfma.hip): S^T = K.Q^T, dP^T = V.dO^T
    constexpr float kExp2 = 0.125f * 1.4426950408889634f;
    const float Lk = Lq * 1.4426950408889634f;
    f32x4 ds[NKT], dp[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      ds[kt] = mfma(*reinterpret_cast<const uint4*>(Ks + (kt * 16 + c) * KS + g * 8), q0, f32x4{0.f, 0.f, 0.f, 0.f});
      dp[kt] = mfma(*reinterpret_cast<const uint4*>(Vs + (kt * 16 + c) * KS + g * 8), g0, f32x4{0.f, 0.f, 0.f, 0.f});
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      ds[kt] = mfma(*reinterpret_cast<const uint4*>(Ks + (kt * 16 + c) * KS + g * 8 + 32), q1, ds[kt]);
      dp[kt] = mfma(*reinterpret_cast<const uint4*>(Vs + (kt * 16 + c) * KS + g * 8 + 32), g1, dp[kt]);
    }
    // space groups: NKT = ceil(nkeys/16) exactly -> only the last tile holds padded keys; text: causal mask
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __builtin_amdgcn_exp2f(fmaf(ds[kt][r], kExp2, -Lk));
        if (TEXT || kt == NKT - 1) {
          const int key = kt * 16 + g * 4 + r;
          p = (key < nkeys && (!TEXT || key <= qrow)) ? p : 0.f;
        }
        ds[kt][r] = p * (dp[kt][r] - dl);
      }
    }
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (NKT + 1) / 2; ++j) {
      uint4 pa;
      pa.x = pack_bf16x2(ds[2 * j][0], ds[2 * j][1]);
      pa.y = pack_bf16x2(ds[2 * j][2], ds[2 * j][3]);
      constexpr int last = NKT - 1;
      const int j1 = 2 * j + 1 <= last ? 2 * j + 1 : last;
      pa.z = 2 * j + 1 <= last ? pack_bf16x2(ds[j1][0], ds[j1][1]) : 0u;
      pa.w = 2 * j + 1 <= last ? pack_bf16x2(ds[j1][2], ds[j1][3]) : 0u;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint16_t* kp = Kt + (size_t)(dt * 16 + c) * LDK + 2 * j * 16 + g * 4;
        const uint2 lo = *reinterpret_cast<const uint2*>(kp);
        uint2 hi = make_uint2(0, 0);
        if (2 * j + 1 <= last) hi = *reinterpret_cast<const uint2*>(kp + 16);
        o[dt] = mfma(pa, make_uint4(lo.x, lo.y, hi.x, hi.y), o[dt]);
      }
    }
    store_tile_rows(ot, o, 0.125f, lane,
                    [&](int row) { return dqkv + (size_t)b * T * ts + (size_t)(tok0 + qt * 16 + row) * ts + h * 64; },
                    [&](int row) { return qt * 16 + row < N; });
  }
}

// ------------------------------------------------------------------------------------------------------------
// dK / dV kernel
// ------------------------------------------------------------------------------------------------------------
struct DkvGeom {
  int QROWS, LDQ;                         // queries padded to a multiple of 32; transposed row stride
  int qs_off, dos_off, qt_off, dot_off, lse_off, del_off, vec_off, ot_off, total;   // bytes
};

inline DkvGeom dkv_geometry(int N) {
  DkvGeom g{};
  g.QROWS = (N + 31) / 32 * 32;
  g.LDQ = g.QROWS + 8;
  g.qs_off = 0;
  g.dos_off = g.qs_off + g.QROWS * KS * 2;
  g.qt_off = g.dos_off + g.QROWS * KS * 2;
  g.dot_off = g.qt_off + 64 * g.LDQ * 2;
  g.lse_off = g.dot_off + 64 * g.LDQ * 2;
  g.del_off = g.lse_off + g.QROWS * 4;
  g.vec_off = g.del_off + g.QROWS * 4;          // f32: qc[64], doc[64], dqc[64], scalars[8]
  g.ot_off = g.vec_off + (3 * 64 + 8) * 4;
  g.total = g.ot_off + 8 * 16 * OS * 2;
  return g;
}

template <bool TEXT>
__global__ __launch_bounds__(512) void space_bwd_dkv_kernel(const uint16_t* __restrict__ qkv,
                                                            const uint16_t* __restrict__ out,
                                                            const uint16_t* __restrict__ dout,
                                                            const float* __restrict__ lse,
                                                            const float* __restrict__ delta,
                                                            uint16_t* __restrict__ dqkv, float* __restrict__ atom_ws,
                                                            int F, int N, int H, DkvGeom G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* Qs = reinterpret_cast<uint16_t*>(smem + G.qs_off);
  uint16_t* dOs = reinterpret_cast<uint16_t*>(smem + G.dos_off);
  uint16_t* Qt = reinterpret_cast<uint16_t*>(smem + G.qt_off);
  uint16_t* dOt = reinterpret_cast<uint16_t*>(smem + G.dot_off);
  float* lse_s = reinterpret_cast<float*>(smem + G.lse_off);
  float* del_s = reinterpret_cast<float*>(smem + G.del_off);
  float* qc = reinterpret_cast<float*>(smem + G.vec_off);        // raw cls query
  float* doc = qc + 64;                                           // d out of the cls row
  float* dqc = doc + 64;                                          // d cls query accumulator (unscaled)
  float* scal = dqc + 64;                                         // [0] lse_c, [1] delta_c
  uint16_t* Ot = reinterpret_cast<uint16_t*>(smem + G.ot_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % H, f = (blockIdx.x / H) % F, b = blockIdx.x / (H * F);
  const int D = H * 64, T = TEXT ? N : 1 + F * N, nkeys = TEXT ? N : N + 1, tok0 = TEXT ? 0 : 1 + f * N;
  const int LDQ = G.LDQ, QROWS = G.QROWS;
  const size_t ts = (size_t)3 * D;
  const uint16_t* base = qkv + (size_t)b * T * ts + h * 64;
  const uint16_t* dobase = dout + (size_t)b * T * D + h * 64;
  const float* lrow = lse + ((size_t)b * H + h) * T;
  const float* drow = delta + ((size_t)b * H + h) * T;

  stage_rows2<512, 4>(Qs, Qt, [&](int r) { return base + (size_t)(tok0 + r) * ts; },
                      dOs, dOt, [&](int r) { return dobase + (size_t)(tok0 + r) * D; }, LDQ, QROWS, N, tid);
  for (int q = tid; q < QROWS; q += 512) {
    lse_s[q] = q < N ? lrow[tok0 + q] * 1.4426950408889634f : INFINITY;   // in log2 units; padded queries: exp2(-inf) = 0
    del_s[q] = q < N ? drow[tok0 + q] : 0.f;
  }
  if (!TEXT && tid < 64) {
    qc[tid] = bf16_to_f32(base[tid]);
    const float go = bf16_to_f32(dobase[tid]);
    doc[tid] = go;
    dqc[tid] = 0.f;
    const float dsum = wave_sum(go * bf16_to_f32(out[(size_t)b * T * D + h * 64 + tid]));
    if (tid == 0) { scal[0] = lrow[0]; scal[1] = dsum; }
  }
  __syncthreads();

  const int c = lane & 15, g = lane >> 4;
  const float Lc = TEXT ? 0.f : scal[0], dlc = TEXT ? 0.f : scal[1];
  uint16_t* ot = Ot + wave * 16 * OS;
  float dqc_part[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) dqc_part[i] = 0.f;

  const int nkt = (nkeys + 15) / 16;
#pragma unroll 1
  for (int kt = wave; kt < nkt; kt += 8) {
    const int krow = kt * 16 + c;
    uint4 k0 = make_uint4(0, 0, 0, 0), k1 = k0, v0 = k0, v1 = k0;
    if (krow < nkeys) {
      const uint16_t* kp = base + (size_t)(TEXT ? krow : (krow == 0 ? 0 : tok0 + krow - 1)) * ts + D + g * 8;
      k0 = *reinterpret_cast<const uint4*>(kp);
      k1 = *reinterpret_cast<const uint4*>(kp + 32);
      v0 = *reinterpret_cast<const uint4*>(kp + D);
      v1 = *reinterpret_cast<const uint4*>(kp + D + 32);
    }
    f32x4 adk[4], adv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { adk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; adv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int qp = 0; qp < QROWS / 32; ++qp) {
      uint4 pa, da;     // A fragments: P^T and dS^T of 32 queries x this key tile
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qt = 2 * qp + t;
        const uint16_t* qsp = Qs + (qt * 16 + c) * KS + g * 8;
        const uint16_t* dsp = dOs + (qt * 16 + c) * KS + g * 8;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = mfma(*reinterpret_cast<const uint4*>(qsp), k0, s);
        s = mfma(*reinterpret_cast<const uint4*>(qsp + 32), k1, s);
        dp = mfma(*reinterpret_cast<const uint4*>(dsp), v0, dp);
        dp = mfma(*reinterpret_cast<const uint4*>(dsp + 32), v1, dp);
        float p[4], d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qt * 16 + g * 4 + r;
          p[r] = (!TEXT || q >= krow) ? __builtin_amdgcn_exp2f(fmaf(s[r], 0.125f * 1.4426950408889634f, -lse_s[q])) : 0.f;
          d[r] = p[r] * (dp[r] - del_s[q]);
        }
        if (t == 0) {
          pa.x = pack_bf16x2(p[0], p[1]); pa.y = pack_bf16x2(p[2], p[3]);
          da.x = pack_bf16x2(d[0], d[1]); da.y = pack_bf16x2(d[2], d[3]);
        } else {
          pa.z = pack_bf16x2(p[0], p[1]); pa.w = pack_bf16x2(p[2], p[3]);
          da.z = pack_bf16x2(d[0], d[1]); da.w = pack_bf16x2(d[2], d[3]);
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint16_t* gp = dOt + (size_t)(dt * 16 + c) * LDQ + qp * 32 + g * 4;
        const uint16_t* qq = Qt + (size_t)(dt * 16 + c) * LDQ + qp * 32 + g * 4;
        const uint2 g_lo = *reinterpret_cast<const uint2*>(gp), g_hi = *reinterpret_cast<const uint2*>(gp + 16);
        const uint2 q_lo = *reinterpret_cast<const uint2*>(qq), q_hi = *reinterpret_cast<const uint2*>(qq + 16);
        adv[dt] = mfma(pa, make_uint4(g_lo.x, g_lo.y, g_hi.x, g_hi.y), adv[dt]);
        adk[dt] = mfma(da, make_uint4(q_lo.x, q_lo.y, q_hi.x, q_hi.y), adk[dt]);
      }
    }

    if constexpr (TEXT) {
      uint16_t* dkb = dqkv + (size_t)b * T * ts + D + h * 64;
      store_tile_rows(ot, adk, 0.125f, lane, [&](int row) { return dkb + (size_t)(kt * 16 + row) * ts; },
                      [&](int row) { return kt * 16 + row < nkeys; });
      store_tile_rows(ot, adv, 1.0f, lane, [&](int row) { return dkb + D + (size_t)(kt * 16 + row) * ts; },
                      [&](int row) { return kt * 16 + row < nkeys; });
      continue;
    }
    // ---- CLS query (attends to every key): rank-1 terms for this key tile --------------------------------
    float kf[16], vf[16];
    {
      float t8[8];
      Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&k0), t8);
#pragma unroll
      for (int i = 0; i < 8; ++i) kf[i] = t8[i];
      Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&k1), t8);
#pragma unroll
      for (int i = 0; i < 8; ++i) kf[8 + i] = t8[i];
      Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&v0), t8);
#pragma unroll
      for (int i = 0; i < 8; ++i) vf[i] = t8[i];
      Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&v1), t8);
#pragma unroll
      for (int i = 0; i < 8; ++i) vf[8 + i] = t8[i];
    }
    float sc = 0.f, dpc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc = fmaf(qc[g * 8 + i], kf[i], sc);
      sc = fmaf(qc[32 + g * 8 + i], kf[8 + i], sc);
      dpc = fmaf(doc[g * 8 + i], vf[i], dpc);
      dpc = fmaf(doc[32 + g * 8 + i], vf[8 + i], dpc);
    }
    sc += __shfl_xor(sc, 16, 64); sc += __shfl_xor(sc, 32, 64);
    dpc += __shfl_xor(dpc, 16, 64); dpc += __shfl_xor(dpc, 32, 64);
    const bool cls_sees = krow < nkeys && (krow > 0 || f == 0);
    const float pc = cls_sees ? __expf(sc * 0.125f - Lc) : 0.f;
    const float dsc = pc * (dpc - dlc);
#pragma unroll
    for (int i = 0; i < 16; ++i) dqc_part[i] = fmaf(dsc, kf[i], dqc_part[i]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pr = __shfl(pc, g * 4 + r, 64), dsr = __shfl(dsc, g * 4 + r, 64);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        adv[dt][r] = fmaf(pr, doc[dt * 16 + c], adv[dt][r]);
        adk[dt][r] = fmaf(dsr, qc[dt * 16 + c], adk[dt][r]);
      }
    }

    // ---- the cls KEY (row 0 of tile 0) collects gradient from every frame: f32 atomics ----------------------
    if (kt == 0 && g == 0) {
      float* kv0 = atom_ws + ((size_t)b * H + h) * 192 + 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        atomicAdd(kv0 + dt * 16 + c, adk[dt][0] * 0.125f);
        atomicAdd(kv0 + 64 + dt * 16 + c, adv[dt][0]);
      }
    }
    uint16_t* dkb = dqkv + (size_t)b * T * ts + D + h * 64;
    store_tile_rows(ot, adk, 0.125f, lane, [&](int row) { return dkb + (size_t)(tok0 + kt * 16 + row - 1) * ts; },
                    [&](int row) { const int kr = kt * 16 + row; return kr >= 1 && kr < nkeys; });
    store_tile_rows(ot, adv, 1.0f, lane, [&](int row) { return dkb + D + (size_t)(tok0 + kt * 16 + row - 1) * ts; },
                    [&](int row) { const int kr = kt * 16 + row; return kr >= 1 && kr < nkeys; });
  }

  if constexpr (TEXT) return;
  // ---- d(cls query): reduce over the 16 key lanes, then across waves in LDS, one atomic per channel ----------
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float v = dqc_part[i];
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    dqc_part[i] = v;
  }
  if (c == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      atomicAdd(dqc + g * 8 + i, dqc_part[i]);
      atomicAdd(dqc + 32 + g * 8 + i, dqc_part[8 + i]);
    }
  }
  __syncthreads();
  if (tid < 64) atomicAdd(atom_ws + ((size_t)b * H + h) * 192 + tid, dqc[tid] * 0.125f);
}

// dqkv[b, token 0, :] = (d cls q | d cls k | d cls v) from the f32 atomic workspace
__global__ __launch_bounds__(192) void cls_grad_finalize_kernel(const float* __restrict__ atom_ws,
                                                                uint16_t* __restrict__ dqkv, int T, int H) {
  const int h = blockIdx.x % H, b = blockIdx.x / H, t = threadIdx.x;      // t in [0,192): part = t/64
  const int D = H * 64;
  dqkv[(size_t)b * T * 3 * D + (t >> 6) * D + h * 64 + (t & 63)] = f32_to_bf16(atom_ws[((size_t)b * H + h) * 192 + t]);
}

template <int NKT, bool TEXT = false>
int launch_dq(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta, int B,
              int F, int N, int H, hipStream_t st) {
  using L = DqLds<NKT>;
  static_assert(L::total <= 160 * 1024, "LDS per CU");
  (void)hipFuncSetAttribute((const void*)space_bwd_dq_kernel<NKT, TEXT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            L::total);
  hipLaunchKernelGGL((space_bwd_dq_kernel<NKT, TEXT>), dim3((unsigned)(B * F * H)), dim3(512), L::total, st,
                     (const uint16_t*)qkv, (const uint16_t*)out, (const uint16_t*)dout, lse, (uint16_t*)dqkv, delta, F,
                     N, H);
  LVL_CHECK_LAUNCH("space_bwd_dq");
  return LVL_OK;
}

}  // namespace

void lvl_launch_cls_grad_finalize(const float* atom_ws, void* dqkv, int B, int T, int H, hipStream_t st) {
  hipLaunchKernelGGL(cls_grad_finalize_kernel, dim3((unsigned)(B * H)), dim3(192), 0, st, atom_ws, (uint16_t*)dqkv, T, H);
}

bool lvl_space_mfma_bwd_supported(int F, int N) {
  return N >= 1 && N + 1 <= 208 && dkv_geometry(N).QROWS <= 256 && dkv_geometry(N).total <= 160 * 1024 && F <= 64;
}

// ws layout: delta [B*H*T] f32, then atomics [B*H*192] f32 (d cls q | d cls k | d cls v)
int lvl_space_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                       int B, int F, int N, int H, hipStream_t st) {
  const int T = 1 + F * N, nkeys = N + 1;
  float* delta = ws;
  float* atom_ws = ws + (size_t)B * H * T;
  hipError_t e = hipMemsetAsync(atom_ws, 0, (size_t)B * H * 192 * sizeof(float), st);
  if (e != hipSuccess) return lvl_fail(LVL_EHIP, "space_bwd memset: %s", hipGetErrorString(e));
  int rc;
  switch ((nkeys + 15) / 16) {          // exact tile count: the kernel masks only the last key tile
#define SPACE_DQ_CASE(K) case K: rc = launch_dq<K>(qkv, out, dout, lse, dqkv, delta, B, F, N, H, st); break;
    SPACE_DQ_CASE(1) SPACE_DQ_CASE(2) SPACE_DQ_CASE(3) SPACE_DQ_CASE(4) SPACE_DQ_CASE(5) SPACE_DQ_CASE(6) SPACE_DQ_CASE(7)
    SPACE_DQ_CASE(8) SPACE_DQ_CASE(9) SPACE_DQ_CASE(10) SPACE_DQ_CASE(11) SPACE_DQ_CASE(12) SPACE_DQ_CASE(13)
#undef SPACE_DQ_CASE
    default: return lvl_fail(LVL_ENOSYS, "space_mfma_bwd: %d keys per group not supported", nkeys);
  }
  if (rc) return rc;
  const DkvGeom G = dkv_geometry(N);
  (void)hipFuncSetAttribute((const void*)space_bwd_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            G.total);
  hipLaunchKernelGGL(space_bwd_dkv_kernel<false>, dim3((unsigned)(B * F * H)), dim3(512), G.total, st, (const uint16_t*)qkv,
                     (const uint16_t*)out, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, atom_ws, F, N, H, G);
  LVL_CHECK_LAUNCH("space_bwd_dkv");
  hipLaunchKernelGGL(cls_grad_finalize_kernel, dim3((unsigned)(B * H)), dim3(192), 0, st, atom_ws, (uint16_t*)dqkv, T, H);
  LVL_CHECK_LAUNCH("cls_grad_finalize");
  return LVL_OK;
}

bool lvl_text_mfma_bwd_supported(int L) { return L >= 1 && L <= 208 && dkv_geometry(L).total <= 160 * 1024; }

// ws: delta [B*H*L] f32
int lvl_text_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int L, int H, hipStream_t st) {
  int rc;
  if (L <= 64) rc = launch_dq<4, true>(qkv, out, dout, lse, dqkv, ws, B, 1, L, H, st);
  else if (L <= 128) rc = launch_dq<8, true>(qkv, out, dout, lse, dqkv, ws, B, 1, L, H, st);
  else rc = launch_dq<13, true>(qkv, out, dout, lse, dqkv, ws, B, 1, L, H, st);
  if (rc) return rc;
  const DkvGeom G = dkv_geometry(L);
  (void)hipFuncSetAttribute((const void*)space_bwd_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, G.total);
  hipLaunchKernelGGL(space_bwd_dkv_kernel<true>, dim3((unsigned)(B * H)), dim3(512), G.total, st, (const uint16_t*)qkv,
                     (const uint16_t*)out, (const uint16_t*)dout, lse, ws, (uint16_t*)dqkv, nullptr, 1, L, H, G);
  LVL_CHECK_LAUNCH("text_bwd_dkv");
  return LVL_OK;
}
