// Space-mode divided attention backward on the matrix cores (bf16, f32 accumulate), gfx950.
//
// With every key of a (sample, frame, head) group LDS-resident the backward recomputes P = exp(S - lse)
// from the saved row log-sum-exp (no max/sum pass) and needs four contractions:
//     dP = dO V^T (over d)     dS = P o (dP - delta),  delta_q = dO_q . O_q
//     dQ = dS K   (over keys)  dK = dS^T Q (over queries)   dV = P^T dO (over queries)
// A 16x16x32 MFMA wants its contraction index contiguous per lane. Contractions over d read 16-byte fragments
// of the row-major LDS images; contractions over keys / queries read the SAME images with the LDS transpose
// read (ds_read_b64_tr_b16, attn_mfma_common.h) as B operands, while the A operand is the freshly computed
// tile itself: the C layout of S^T (resp. S) puts one query (resp. key) per lane, which is exactly the
// A-fragment layout with a permuted k-order (same trick as the forward's P.V). No transposed copies exist.
// Video groups of up to 288 keys run on the FUSED kernel further down (one kernel, every operand staged once,
// 32 x 32 register blocks, the cls query handled as one more query row). The causal text tower and the large groups
// (289..592 keys, whose K+V or Q+dO images alone fill the LDS) use the two-kernel form, each kernel one workgroup
// per (b, f, h):
//   dq kernel : waves own 16-query tiles, all keys resident (K, V images) -> dQ, delta
//   dkv kernel: waves own 16-key tiles, all queries resident (Q, dO images) -> dK, dV; also folds in the CLS
//               query's rank-1 contributions to dK/dV (it attends to every key, timesformer.py:116-119) and
//               accumulates d(cls q) and d(cls k,v) -- which receive gradient from every frame -- with f32
//               atomics into a workspace finalised by a tiny kernel.
// PRECISION POLICY (template P, attn_mfma_common.h): PrecBf16 = the benched kernels; PrecSplit = the same kernel text on
// float32 tensors, operands as hi/lo bf16 images, 3 MFMAs per product (f32-class: what the parity configuration runs).
#include "attn_mfma_common.h"

// SBW_PACKED 1: the fused kernel's score arithmetic on v_pk_fma / v_pk_mul_f32 pairs (default), 0: scalar -- measured 0.7 % SLOWER
// in round 6 (0.8302 -> 0.8364 ms, profiles/r06_valu_rates.txt): here the pairs feed two transcendentals each and the
// packed forms win, unlike the SLP vectoriser's pairs elsewhere in this file (-fno-slp-vectorize: -3.2 %)
#ifndef SBW_PACKED
#define SBW_PACKED 1
#endif

namespace {

using namespace attn_mfma;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kExp2 = 0.125f * kLog2e;          // exp(s * scale) = exp2(s * kExp2)

// ------------------------------------------------------------------------------------------------------------
// dQ kernel
// ------------------------------------------------------------------------------------------------------------
template <int NKT, int NW, int IMAGES = 1> struct DqLds {
  static constexpr int KROWS = NKT * 16;
  static constexpr int ks_off = 0;
  static constexpr int vs_off = ks_off + KROWS * RS * 2;
  static constexpr int lo_off = 2 * KROWS * RS;                 // elements, hi image -> lo image (PrecSplit)
  static constexpr int ot_off = IMAGES * (vs_off + KROWS * RS * 2);
  static constexpr int total = ot_off + NW * 16 * OS * 2;
};

// NW = 8 waves (two workgroups per CU) up to 272 keys; NW = 4 (one workgroup per CU, one wave per SIMD) for the
// large groups whose K and V images fill the LDS (TSF-L/14 at 336: 577 keys). MASKALL: NKT is an upper bound of the
// tile count, every tile is masked against nkeys.
template <typename P, int NKT, bool TEXT, int NW, bool MASKALL>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 1 : ((NKT <= 13 && !P::kSplit) ? 4 : 2))) void space_bwd_dq_kernel(
    const typename P::io_t* __restrict__ qkv, const typename P::io_t* __restrict__ out,
    const typename P::io_t* __restrict__ dout, const float* __restrict__ lse, typename P::io_t* __restrict__ dqkv,
    float* __restrict__ delta, int F, int N, int H) {
  using io_t = typename P::io_t;
  using Op = typename P::Op;
  using Tr = typename P::Tr;
  constexpr int NT = NW * 64;
  using L = DqLds<NKT, NW, P::kImages>;
  constexpr int LO = L::lo_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* Ks = reinterpret_cast<uint16_t*>(smem + L::ks_off);
  uint16_t* Vs = reinterpret_cast<uint16_t*>(smem + L::vs_off);
  uint16_t* Ot = reinterpret_cast<uint16_t*>(smem + L::ot_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % H, f = (blockIdx.x / H) % F, b = blockIdx.x / (H * F);
  const int D = H * 64, T = TEXT ? N : 1 + F * N, nkeys = TEXT ? N : N + 1, tok0 = TEXT ? 0 : 1 + f * N;
  const size_t ts = (size_t)3 * D;
  const io_t* base = qkv + (size_t)b * T * ts + h * 64;
  const io_t* obase = out + (size_t)b * T * D + h * 64;
  const io_t* dobase = dout + (size_t)b * T * D + h * 64;

  // fragments of this wave's first query tile: issued before the staging so that both are in flight together
  const int c = lane & 15, g = lane >> 4;
  auto tok_of = [&](int qt) { const int qr = qt * 16 + c; return tok0 + (qr < N ? qr : N - 1); };
  Op nq0, nq1, ng0, ng1, ny0, ny1;
  auto load_frags = [&](int qt) {
    const int tk = tok_of(qt);
    const io_t* qp = base + (size_t)tk * ts + g * 8;
    nq0 = P::load_op(qp); nq1 = P::load_op(qp + 32);
    ng0 = P::load_op(dobase + (size_t)tk * D + g * 8);
    ng1 = P::load_op(dobase + (size_t)tk * D + g * 8 + 32);
    ny0 = P::load_op(obase + (size_t)tk * D + g * 8);
    ny1 = P::load_op(obase + (size_t)tk * D + g * 8 + 32);
  };
  load_frags(wave * 16 < N ? wave : 0);

  {   // key row r = token tok0 + r - 1 (r >= 1) or the cls token (r = 0); text: token r
    const io_t* krow0 = base + (size_t)(TEXT ? 0 : tok0 - 1) * ts + D;
    constexpr int RPP = NT / 8, GROUP = 8 * RPP;        // at most 8 passes (16 loads per thread) in flight
    constexpr int MAXP = (L::KROWS < GROUP ? L::KROWS + RPP - 1 : GROUP) / RPP;
#pragma unroll 1
    for (int r0 = 0; r0 < L::KROWS; r0 += GROUP) {
      const int pad = L::KROWS - r0 < GROUP ? L::KROWS - r0 : GROUP;
      stage_rows2<P, NT, MAXP>(Ks + r0 * RS, krow0 + (size_t)r0 * ts, ts, (TEXT || r0 != 0) ? nullptr : base + D,
                               Vs + r0 * RS, krow0 + D + (size_t)r0 * ts, ts,
                               (TEXT || r0 != 0) ? nullptr : base + 2 * D, pad, nkeys - r0, tid, LO);
    }
  }
  __syncthreads();

  uint16_t* ot = Ot + wave * 16 * OS;
  const FragOff fo = frag_offsets(lane);
#pragma unroll 1
  for (int qt = wave; qt * 16 < N; qt += NW) {
    const int qrow = qt * 16 + c;
    const int tok = tok_of(qt);
    const Op q0 = nq0, q1 = nq1, g0 = ng0, g1 = ng1;
    float dl;
    {
      float a[8], bb[8];
      P::to_f32(ng0, a);
      P::to_f32(ny0, bb);
      dl = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) dl = fmaf(a[i], bb[i], dl);
      P::to_f32(ng1, a);
      P::to_f32(ny1, bb);
#pragma unroll
      for (int i = 0; i < 8; ++i) dl = fmaf(a[i], bb[i], dl);
      dl = rows4_sum(dl);
    }
    if ((qt + NW) * 16 < N) load_frags(qt + NW);
    const size_t srow = ((size_t)b * H + h) * T + tok;
    const float Lk = lse[srow] * kLog2e;
    if (g == 0 && qrow < N) delta[srow] = dl;

    // Key tiles in pairs: S^T = K.Q^T and dP^T = V.dO^T (four independent accumulate chains per pair), then
    // dS^T packed straight into the A fragment of the dQ contraction -- the f32 tiles never pile up.
    Op pa[(NKT + 1) / 2];
#pragma unroll
    for (int j = 0; j < (NKT + 1) / 2; ++j) {
      constexpr int last = NKT - 1;
      const int k0 = 2 * j, k1 = 2 * j + 1 <= last ? 2 * j + 1 : last;
      const bool two = 2 * j + 1 <= last;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
      s0 = mfma(P::tile_op(Ks, LO, k0, fo.a[0]), q0, s0);
      p0 = mfma(P::tile_op(Vs, LO, k0, fo.a[0]), g0, p0);
      if (two) {
        s1 = mfma(P::tile_op(Ks, LO, k1, fo.a[0]), q0, s1);
        p1 = mfma(P::tile_op(Vs, LO, k1, fo.a[0]), g0, p1);
      }
      s0 = mfma(P::tile_op(Ks, LO, k0, fo.a[1]), q1, s0);
      p0 = mfma(P::tile_op(Vs, LO, k0, fo.a[1]), g1, p0);
      if (two) {
        s1 = mfma(P::tile_op(Ks, LO, k1, fo.a[1]), q1, s1);
        p1 = mfma(P::tile_op(Vs, LO, k1, fo.a[1]), g1, p1);
      }
      float d0[4], d1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e0 = __builtin_amdgcn_exp2f(fmaf(s0[r], kExp2, -Lk));
        float e1 = __builtin_amdgcn_exp2f(fmaf(s1[r], kExp2, -Lk));
        // space groups: NKT = ceil(nkeys/16) exactly -> only the last tile holds padded keys; text: causal
        if (TEXT || MASKALL || k0 == last) {
          const int key = k0 * 16 + g * 4 + r;
          e0 = (key < nkeys && (!TEXT || key <= qrow)) ? e0 : 0.f;
        }
        if (TEXT || MASKALL || k1 == last) {
          const int key = k1 * 16 + g * 4 + r;
          e1 = (key < nkeys && (!TEXT || key <= qrow)) ? e1 : 0.f;
        }
        d0[r] = e0 * (p0[r] - dl);
        d1[r] = two ? e1 * (p1[r] - dl) : 0.f;
      }
      pa[j] = P::pack(d0, d1);
    }
    // dQ = dS . K: B fragments are transpose reads of the K image (4 consecutive keys per half)
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (NKT + 1) / 2; ++j) {
      constexpr int last = NKT - 1;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const Tr lo = P::tile_tr(Ks, LO, 2 * j, fo.tr[dt]);
        Tr hi = P::zero_tr();
        if (2 * j + 1 <= last) hi = P::tile_tr(Ks, LO, 2 * j + 1, fo.tr[dt]);
        o[dt] = mfma(pa[j], P::join(lo, hi), o[dt]);
      }
    }
    store_tile_rows<P>(ot, o, 0.125f, lane,
                    [&](int row) { return dqkv + (size_t)b * T * ts + (size_t)(tok0 + qt * 16 + row) * ts + h * 64; },
                    [&](int row) { return qt * 16 + row < N; });
  }
}

// ------------------------------------------------------------------------------------------------------------
// dK / dV kernel
// ------------------------------------------------------------------------------------------------------------
struct DkvGeom {
  int QROWS;                                                     // queries padded to a multiple of 32
  int qs_off, dos_off, lse_off, del_off, vec_off, ot_off, total;  // bytes
  int lo_off;                                                    // ELEMENTS from a hi image to its lo image (PrecSplit)
};

inline DkvGeom dkv_geometry(int N, int nw = 8, int images = 1) {
  DkvGeom g{};
  g.QROWS = (N + 31) / 32 * 32;
  g.qs_off = 0;
  g.dos_off = g.qs_off + g.QROWS * RS * 2;
  g.lo_off = 2 * g.QROWS * RS;
  g.lse_off = images * (g.dos_off + g.QROWS * RS * 2);
  g.del_off = g.lse_off + g.QROWS * 4;
  g.vec_off = g.del_off + g.QROWS * 4;          // f32: qc[64], doc[64], dqc[64], scalars[8]
  g.ot_off = g.vec_off + (3 * 64 + 8) * 4;
  g.total = g.ot_off + nw * 16 * OS * 2;
  return g;
}

template <typename P, bool TEXT, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 1 : (P::kSplit ? 2 : 4))) void space_bwd_dkv_kernel(
    const typename P::io_t* __restrict__ qkv, const typename P::io_t* __restrict__ out,
    const typename P::io_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
    typename P::io_t* __restrict__ dqkv, float* __restrict__ atom_ws, int F, int N, int H, DkvGeom G) {
  using io_t = typename P::io_t;
  using Op = typename P::Op;
  using Tr = typename P::Tr;
  const int LO = G.lo_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* Qs = reinterpret_cast<uint16_t*>(smem + G.qs_off);
  uint16_t* dOs = reinterpret_cast<uint16_t*>(smem + G.dos_off);
  float* lse_s = reinterpret_cast<float*>(smem + G.lse_off);
  float* del_s = reinterpret_cast<float*>(smem + G.del_off);
  float* qc = reinterpret_cast<float*>(smem + G.vec_off);        // raw cls query
  float* doc = qc + 64;                                           // d out of the cls row
  float* dqc = doc + 64;                                          // d cls query accumulator (unscaled)
  float* scal = dqc + 64;                                         // [0] lse_c (log2 units), [1] delta_c
  uint16_t* Ot = reinterpret_cast<uint16_t*>(smem + G.ot_off);

  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % H, f = (blockIdx.x / H) % F, b = blockIdx.x / (H * F);
  const int D = H * 64, T = TEXT ? N : 1 + F * N, nkeys = TEXT ? N : N + 1, tok0 = TEXT ? 0 : 1 + f * N;
  const int QROWS = G.QROWS;
  const size_t ts = (size_t)3 * D;
  const io_t* base = qkv + (size_t)b * T * ts + h * 64;
  const io_t* dobase = dout + (size_t)b * T * D + h * 64;
  const float* lrow = lse + ((size_t)b * H + h) * T;
  const float* drow = delta + ((size_t)b * H + h) * T;
  const int c = lane & 15, g = lane >> 4;
  const int nkt = (nkeys + 15) / 16;

  // K/V fragments of this wave's first key tile: issued before the staging
  Op nk0, nk1, nv0, nv1;
  auto load_kv = [&](int kt) {
    const int krow = kt * 16 + c;
    nk0 = P::zero_op(); nk1 = nk0; nv0 = nk0; nv1 = nk0;
    if (krow < nkeys) {
      const io_t* kp = base + (size_t)(TEXT ? krow : (krow == 0 ? 0 : tok0 + krow - 1)) * ts + D + g * 8;
      nk0 = P::load_op(kp);
      nk1 = P::load_op(kp + 32);
      nv0 = P::load_op(kp + D);
      nv1 = P::load_op(kp + D + 32);
    }
  };
  load_kv(wave < nkt ? wave : 0);

  {
    constexpr int GROUP = 4 * (NT / 8);          // 4 passes (8 loads per thread) in flight at a time
#pragma unroll 1
    for (int r0 = 0; r0 < QROWS; r0 += GROUP) {
      const int pad = QROWS - r0 < GROUP ? QROWS - r0 : GROUP;
      stage_rows2<P, NT, 4>(Qs + r0 * RS, base + (size_t)(tok0 + r0) * ts, ts, nullptr, dOs + r0 * RS,
                            dobase + (size_t)(tok0 + r0) * D, (size_t)D, nullptr, pad, N - r0, tid, LO);
    }
  }
  for (int q = tid; q < QROWS; q += NT) {
    lse_s[q] = q < N ? lrow[tok0 + q] * kLog2e : INFINITY;      // log2 units; padded queries: exp2(-inf) = 0
    del_s[q] = q < N ? drow[tok0 + q] : 0.f;
  }
  if (!TEXT && tid < 64) {
    qc[tid] = P::to_f32_1(base[tid]);
    const float go = P::to_f32_1(dobase[tid]);
    doc[tid] = go;
    dqc[tid] = 0.f;
    const float dsum = wave_sum(go * P::to_f32_1(out[(size_t)b * T * D + h * 64 + tid]));
    if (tid == 0) { scal[0] = lrow[0] * kLog2e; scal[1] = dsum; }
  }
  __syncthreads();

  const float Lc = TEXT ? 0.f : scal[0], dlc = TEXT ? 0.f : scal[1];
  uint16_t* ot = Ot + wave * 16 * OS;
  const FragOff fo = frag_offsets(lane);

#pragma unroll 1
  for (int kt = wave; kt < nkt; kt += NW) {
    const int krow = kt * 16 + c;
    if (kt != wave) load_kv(kt);          // first tile's fragments were loaded before the staging
    const Op k0 = nk0, k1 = nk1, v0 = nv0, v1 = nv1;
    f32x4 adk[4], adv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { adk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; adv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int qp = 0; qp < QROWS / 32; ++qp) {
      // S = Q.K^T and dP = dO.V^T for 32 queries x this key tile: four independent accumulate chains
      const uint16_t* Qp = Qs + qp * 32 * RS;         // 32-query slab of the two images
      const uint16_t* Gp = dOs + qp * 32 * RS;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, p0 = s0, p1 = s0;
      s0 = mfma(P::tile_op(Qp, LO, 0, fo.a[0]), k0, s0);
      s1 = mfma(P::tile_op(Qp, LO, 1, fo.a[0]), k0, s1);
      p0 = mfma(P::tile_op(Gp, LO, 0, fo.a[0]), v0, p0);
      p1 = mfma(P::tile_op(Gp, LO, 1, fo.a[0]), v0, p1);
      s0 = mfma(P::tile_op(Qp, LO, 0, fo.a[1]), k1, s0);
      s1 = mfma(P::tile_op(Qp, LO, 1, fo.a[1]), k1, s1);
      p0 = mfma(P::tile_op(Gp, LO, 0, fo.a[1]), v1, p0);
      p1 = mfma(P::tile_op(Gp, LO, 1, fo.a[1]), v1, p1);
      float e0[4], e1[4], d0[4], d1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q0i = qp * 32 + g * 4 + r, q1i = q0i + 16;
        e0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], kExp2, -lse_s[q0i]));
        e1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], kExp2, -lse_s[q1i]));
        if (TEXT) {                                   // causal: query sees key iff q >= key
          e0[r] = q0i >= krow ? e0[r] : 0.f;
          e1[r] = q1i >= krow ? e1[r] : 0.f;
        }
        d0[r] = e0[r] * (p0[r] - del_s[q0i]);
        d1[r] = e1[r] * (p1[r] - del_s[q1i]);
      }
      const Op pa = P::pack(e0, e1);
      const Op da = P::pack(d0, d1);
      // dV += P^T dO, dK += dS^T Q: B fragments = transpose reads (4 consecutive queries per half)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const Tr g_lo = P::tile_tr(Gp, LO, 0, fo.tr[dt]), g_hi = P::tile_tr(Gp, LO, 1, fo.tr[dt]);
        const Tr q_lo = P::tile_tr(Qp, LO, 0, fo.tr[dt]), q_hi = P::tile_tr(Qp, LO, 1, fo.tr[dt]);
        adv[dt] = mfma(pa, P::join(g_lo, g_hi), adv[dt]);
        adk[dt] = mfma(da, P::join(q_lo, q_hi), adk[dt]);
      }
    }

    io_t* dkb = dqkv + (size_t)b * T * ts + D + h * 64;
    if constexpr (TEXT) {
      store_tile_rows<P>(ot, adk, 0.125f, lane, [&](int row) { return dkb + (size_t)(kt * 16 + row) * ts; },
                         [&](int row) { return kt * 16 + row < nkeys; });
      store_tile_rows<P>(ot, adv, 1.0f, lane, [&](int row) { return dkb + D + (size_t)(kt * 16 + row) * ts; },
                         [&](int row) { return kt * 16 + row < nkeys; });
      continue;
    }
    // ---- CLS query (attends to every key): rank-1 terms for this key tile --------------------------------
    // (fragments are unpacked half by half to keep the register footprint of this section small)
    float sc = 0.f, dpc = 0.f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float kf[8], vf[8];
      P::to_f32(hh ? k1 : k0, kf);
      P::to_f32(hh ? v1 : v0, vf);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sc = fmaf(qc[hh * 32 + g * 8 + i], kf[i], sc);
        dpc = fmaf(doc[hh * 32 + g * 8 + i], vf[i], dpc);
      }
    }
    sc = rows4_sum(sc);
    dpc = rows4_sum(dpc);
    const bool cls_sees = krow < nkeys && (krow > 0 || f == 0);
    const float pc = cls_sees ? __builtin_amdgcn_exp2f(fmaf(sc, kExp2, -Lc)) : 0.f;
    const float dsc = pc * (dpc - dlc);
    // d(cls q) += dsc * K[key][:]: reduce over the 16 key lanes, then into the workgroup accumulator in LDS
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float kf[8];
      P::to_f32(hh ? k1 : k0, kf);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v = dsc * kf[i];
        v = row16_sum(v);           // DPP row reduction over the 16 key lanes (no LDS crossbar)
        if (c == 0) atomicAdd(dqc + hh * 32 + g * 8 + i, v);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pr = __shfl(pc, g * 4 + r, 64), dsr = __shfl(dsc, g * 4 + r, 64);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        adv[dt][r] = fmaf(pr, doc[dt * 16 + c], adv[dt][r]);
        adk[dt][r] = fmaf(dsr, qc[dt * 16 + c], adk[dt][r]);
      }
    }

    // ---- the cls KEY (row 0 of tile 0) collects gradient from every frame: this frame's share goes to slot f of the
    // partial slab (one writer per slot; cls_grad_finalize_kernel adds the slots up in order: no atomics, deterministic)
    if (kt == 0 && g == 0) {
      float* kv0 = atom_ws + (((size_t)b * H + h) * F + f) * 192 + 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        kv0[dt * 16 + c] = adk[dt][0] * 0.125f;
        kv0[64 + dt * 16 + c] = adv[dt][0];
      }
    }
    store_tile_rows<P>(ot, adk, 0.125f, lane, [&](int row) { return dkb + (size_t)(tok0 + kt * 16 + row - 1) * ts; },
                       [&](int row) { const int kr = kt * 16 + row; return kr >= 1 && kr < nkeys; });
    store_tile_rows<P>(ot, adv, 1.0f, lane, [&](int row) { return dkb + D + (size_t)(tok0 + kt * 16 + row - 1) * ts; },
                       [&](int row) { const int kr = kt * 16 + row; return kr >= 1 && kr < nkeys; });
  }

  if constexpr (TEXT) return;
  // ---- d(cls query): the per-tile sums were accumulated in LDS; this frame's share goes to slot f --------------
  __syncthreads();
  if (tid < 64) atom_ws[(((size_t)b * H + h) * F + f) * 192 + tid] = dqc[tid] * 0.125f;
}

// ------------------------------------------------------------------------------------------------------------
// Fused backward of the video groups: ONE kernel stages every operand once
// ------------------------------------------------------------------------------------------------------------
// The two kernels above read q, k, v, dO from HBM twice and block 16 keys (queries) per wave, so every Q/dO (K/V)
// fragment is re-read from LDS for every tile; the cls query is folded in by a separate scalar section. Here a
// workgroup of 4 waves (two workgroups per CU, <= 256 VGPRs) runs both passes over ONE pair of LDS images, blocks
// 32 x 32, and treats the cls query as query row N of the group (it attends to the frame's patch keys, and to the cls
// key in frame 0 only -- timesformer.py:116-119 -- with its own global lse):
//   phase 1  images = K, V. A wave owns 32 queries (fragments straight from global): every K/V fragment read feeds
//            two query tiles; dQ is accumulated as soon as a key pair's dS is packed; delta and lse of the group's
//            queries stay in LDS for phase 2. The partial dQ of the cls row goes to the f32 atomic slab.
//   phase 2  images = Q, dO (re-staged over K, V; the loads fly while the slower waves finish phase 1). A wave owns
//            32 keys: every Q/dO fragment and every transpose read feeds two key tiles.
// All gradient tiles are accumulated TRANSPOSED (channels x rows: the weight-like operand goes first), so a lane ends
// up with 4 consecutive channels of one token and stores them directly -- no LDS transposition of the results.
template <int NKP, int IMAGES = 1> struct FusedLds {
  static constexpr int R = NKP * 32;                       // image rows: keys padded to whole pairs of tiles
  static constexpr int img0_off = 0;
  static constexpr int img1_off = R * RS * 2;
  static constexpr int lo_off = 2 * R * RS;                // ELEMENTS from a hi image to its lo image (PrecSplit)
  static constexpr int lse_off = IMAGES * 2 * R * RS * 2;
  static constexpr int del_off = lse_off + R * 4;
  static constexpr int qsum_off = del_off + R * 4;         // [4 waves][64 channels] f32: column sums of dQ (bias-gradient rider)
  static constexpr int total = qsum_off + 4 * 64 * 4;
};

// sum over the 16 lanes of a row (the 16 tokens of an accumulator tile), result in every lane: four DPP row rotations
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));   // row_ror:8
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));   // row_ror:4
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));   // row_ror:2
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));   // row_ror:1
  return x;
}

// o[dt][r] = X^T[channel dt*16 + g*4 + r][token lane&15] -> 4 stores of 4 consecutive channels of one token row
template <typename P>
__device__ __forceinline__ void store_token_channels(typename P::io_t* row, const f32x4 (&o)[4], float mul, int g) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    P::store4(row + dt * 16 + g * 4, o[dt][0] * mul, o[dt][1] * mul, o[dt][2] * mul, o[dt][3] * mul);
}
// the same into one slot of the f32 partial slab of the cls token's gradients (a slot part has ONE writer)
__device__ __forceinline__ void slab_token_channels(float* dst, const f32x4 (&o)[4], float mul, int g) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    *reinterpret_cast<float4*>(dst + dt * 16 + g * 4) = make_float4(o[dt][0] * mul, o[dt][1] * mul, o[dt][2] * mul, o[dt][3] * mul);
}

template <typename P, int NKP>
__global__ __launch_bounds__(256, (P::kSplit ? 1 : 2)) void space_bwd_fused_kernel(
    const typename P::io_t* __restrict__ qkv, const typename P::io_t* __restrict__ out,
    const typename P::io_t* __restrict__ dout, const float* __restrict__ lse, typename P::io_t* __restrict__ dqkv,
    float* __restrict__ atom_ws, float* __restrict__ dq_part, int F, int N, int H) {
  // dq_part (nullable; round 5): [B * F, H * 64] f32 -- this workgroup's column sums of dQ over its N patch queries and
  // its share of the cls query's dQ, i.e. its part of the q third of d(qkv bias). The accumulators are already in
  // registers: 32 adds per query pair, 64 DPP adds and one 256-byte store per workgroup instead of a pass over dqkv.
  using io_t = typename P::io_t;
  using Op = typename P::Op;
  using Tr = typename P::Tr;
  constexpr int NW = 4, NT = 256, R = NKP * 32;
  using L = FusedLds<NKP, P::kImages>;
  constexpr int LO = L::lo_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* img0 = reinterpret_cast<uint16_t*>(smem + L::img0_off);      // K, then Q
  uint16_t* img1 = reinterpret_cast<uint16_t*>(smem + L::img1_off);      // V, then dO
  float* lse_s = reinterpret_cast<float*>(smem + L::lse_off);            // log2 units
  float* del_s = reinterpret_cast<float*>(smem + L::del_off);
  float* qsum_s = reinterpret_cast<float*>(smem + L::qsum_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % H, f = (blockIdx.x / H) % F, b = blockIdx.x / (H * F);
  const int D = H * 64, T = 1 + F * N, nkeys = N + 1, tok0 = 1 + f * N;
  const int nqp = (N + 1 + 31) / 32;                    // query pairs: N patch queries + the cls query (row N)
  const size_t ts = (size_t)3 * D;
  const io_t* base = qkv + (size_t)b * T * ts + h * 64;
  const io_t* obase = out + (size_t)b * T * D + h * 64;
  const io_t* dobase = dout + (size_t)b * T * D + h * 64;
  const float* lrow = lse + ((size_t)b * H + h) * T;
  float* cls_ws = atom_ws + (((size_t)b * H + h) * F + f) * 192;  // this frame's slot: d cls q | d cls k | d cls v
  const int c = lane & 15, g = lane >> 4;
  const FragOff fo = frag_offsets(lane);
  // token of query row qr: patch rows, then the cls token; padding rows alias a valid token (never stored)
  auto tok_of_row = [&](int qr) { return qr < N ? tok0 + qr : (qr == N ? 0 : tok0); };

  // ---- phase 1: dQ (and delta) --------------------------------------------------------------------------------
  Op qf[2][2], gf[2][2], yf[2][2];
  float lraw[2];
  auto load_qfrags = [&](int qp) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int tk = tok_of_row((2 * qp + t) * 16 + c);
      lraw[t] = lrow[tk];
      const io_t* qptr = base + (size_t)tk * ts + g * 8;
      qf[t][0] = P::load_op(qptr);
      qf[t][1] = P::load_op(qptr + 32);
      gf[t][0] = P::load_op(dobase + (size_t)tk * D + g * 8);
      gf[t][1] = P::load_op(dobase + (size_t)tk * D + g * 8 + 32);
      yf[t][0] = P::load_op(obase + (size_t)tk * D + g * 8);
      yf[t][1] = P::load_op(obase + (size_t)tk * D + g * 8 + 32);
    }
  };
  load_qfrags(wave < nqp ? wave : 0);                    // in flight together with the staging

  for (int i = tid; i < R; i += NT) { lse_s[i] = INFINITY; del_s[i] = 0.f; }      // padded queries: exp2(-inf) = 0
  {   // key row r = token tok0 + r - 1 (r >= 1) or the cls token (r = 0)
    const io_t* krow0 = base + (size_t)(tok0 - 1) * ts + D;
    stage_rows2<P, NT, NKP>(img0, krow0, ts, base + D, img1, krow0 + D, ts, base + 2 * D, R, nkeys, tid, LO);
  }
  __syncthreads();

  f32x4 qsum[4];                                         // column sums of dQ^T over this wave's queries (rider)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) qsum[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int qp = wave; qp < nqp; qp += NW) {
    if (qp != wave) load_qfrags(qp);
    float dl[2], Lk[2];
    bool kill0[2];                                       // (cls query, cls key) outside frame 0: not attended
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a[8], bb[8], acc = 0.f;
      P::to_f32(gf[t][0], a);
      P::to_f32(yf[t][0], bb);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(a[i], bb[i], acc);
      P::to_f32(gf[t][1], a);
      P::to_f32(yf[t][1], bb);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(a[i], bb[i], acc);
      acc = rows4_sum(acc);
      dl[t] = acc;
      const int qrow = (2 * qp + t) * 16 + c;
      Lk[t] = lraw[t] * kLog2e;
      kill0[t] = f != 0 && qrow == N && g == 0;
      if (g == 0 && qrow <= N) { del_s[qrow] = acc; lse_s[qrow] = Lk[t]; }
    }
    f32x4 o[2][4];                                       // dQ^T: [channel dt*16 + g*4 + r][query c]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int j = 0; j < NKP; ++j) {
      // S^T = K.Q^T and dP^T = V.dO^T for key tiles 2j, 2j+1 x query tiles 0, 1: each K/V fragment feeds two MFMAs
      Op kf[2][2], vf[2][2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          kf[kt][hh] = P::tile_op(img0, LO, 2 * j + kt, fo.a[hh]);
          vf[kt][hh] = P::tile_op(img1, LO, 2 * j + kt, fo.a[hh]);
        }
      Op pa[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        f32x4 p0 = {-dl[t], -dl[t], -dl[t], -dl[t]}, p1 = p0;     // dP - delta: delta rides in the accumulator
        s0 = mfma(kf[0][0], qf[t][0], s0);
        p0 = mfma(vf[0][0], gf[t][0], p0);
        s1 = mfma(kf[1][0], qf[t][0], s1);
        p1 = mfma(vf[1][0], gf[t][0], p1);
        s0 = mfma(kf[0][1], qf[t][1], s0);
        p0 = mfma(vf[0][1], gf[t][1], p0);
        s1 = mfma(kf[1][1], qf[t][1], s1);
        p1 = mfma(vf[1][1], gf[t][1], p1);
        // score pairs: v_pk_fma_f32 for the exponent, v_pk_mul_f32 for dS = P (dP - delta)
        float d0[4], d1[4];
#if SBW_PACKED
        const f32x2 k2 = {kExp2, kExp2}, nl2 = {-Lk[t], -Lk[t]};
#endif
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
#if SBW_PACKED
          const f32x2 a0 = f32x2{s0[r], s0[r + 1]} * k2 + nl2, a1 = f32x2{s1[r], s1[r + 1]} * k2 + nl2;
#else
          const float a0[2] = {fmaf(s0[r], kExp2, -Lk[t]), fmaf(s0[r + 1], kExp2, -Lk[t])};
          const float a1[2] = {fmaf(s1[r], kExp2, -Lk[t]), fmaf(s1[r + 1], kExp2, -Lk[t])};
#endif
          f32x2 e0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
          f32x2 e1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
          if (j == 0 && r == 0) e0[0] = kill0[t] ? 0.f : e0[0];
          if (j == NKP - 1) {            // the padded key rows are zero, but exp2(-lse) may overflow: mask them
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              e0[u] = (2 * j) * 16 + g * 4 + r + u < nkeys ? e0[u] : 0.f;
              e1[u] = (2 * j + 1) * 16 + g * 4 + r + u < nkeys ? e1[u] : 0.f;
            }
          }
#if SBW_PACKED
          const f32x2 x0 = e0 * f32x2{p0[r], p0[r + 1]}, x1 = e1 * f32x2{p1[r], p1[r + 1]};
          d0[r] = x0[0]; d0[r + 1] = x0[1];
          d1[r] = x1[0]; d1[r + 1] = x1[1];
#else
          d0[r] = e0[0] * p0[r]; d0[r + 1] = e0[1] * p0[r + 1];
          d1[r] = e1[0] * p1[r]; d1[r + 1] = e1[1] * p1[r + 1];
#endif
        }
        pa[t] = P::pack(d0, d1);
      }
      // dQ^T += K^T . dS^T for the 32 keys of this pair: A fragments are transpose reads of the K image
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const Tr lo = P::tile_tr(img0, LO, 2 * j, fo.tr[dt]);
        const Tr hi = P::tile_tr(img0, LO, 2 * j + 1, fo.tr[dt]);
        const Op kb = P::join(lo, hi);
        o[0][dt] = mfma(kb, pa[0], o[0][dt]);
        o[1][dt] = mfma(kb, pa[1], o[1][dt]);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qrow = (2 * qp + t) * 16 + c;
      if (qrow < N)
        store_token_channels<P>(dqkv + (size_t)b * T * ts + (size_t)(tok0 + qrow) * ts + h * 64, o[t], 0.125f, g);
      else if (qrow == N)
        slab_token_channels(cls_ws, o[t], 0.125f, g);            // this frame's share of d(cls q)
      if (dq_part && qrow <= N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) qsum[dt] += o[t][dt];
      }
    }
  }
  if (dq_part) {                                         // uniform; lanes c == 0 publish the wave's 64 channel sums
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = row16_sum(qsum[dt][r]);
        if (c == 0) qsum_s[wave * 64 + dt * 16 + g * 4 + r] = v;
      }
  }

  // ---- phase 2: dK, dV ----------------------------------------------------------------------------------------
  Op kk[2][2], vv[2][2];
  auto load_kv = [&](int kp) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int krow = (2 * kp + t) * 16 + c;
      kk[t][0] = P::zero_op(); kk[t][1] = kk[t][0]; vv[t][0] = kk[t][0]; vv[t][1] = kk[t][0];
      if (krow < nkeys) {
        const io_t* kptr = base + (size_t)(krow == 0 ? 0 : tok0 + krow - 1) * ts + D + g * 8;
        kk[t][0] = P::load_op(kptr);
        kk[t][1] = P::load_op(kptr + 32);
        vv[t][0] = P::load_op(kptr + D);
        vv[t][1] = P::load_op(kptr + D + 32);
      }
    }
  };
  // This wave's FIRST key pair comes from the K, V images while they are still in LDS (they are overwritten behind the
  // barrier below): the same fragments as load_kv's (row c of the tile, channels g*8.. and 32 + g*8..; padded rows are
  // zero in the images) without a second trip over the fabric -- 4 of the 7 key pairs of a TSF-B group. The second pair
  // of waves 0..2 is fetched from global memory inside the loop as before (round 6).
  if (wave < NKP) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        kk[t][hh] = P::tile_op(img0, LO, 2 * wave + t, fo.a[hh]);
        vv[t][hh] = P::tile_op(img1, LO, 2 * wave + t, fo.a[hh]);
      }
  }
  {   // Q and dO rows (patch queries, then the cls query as row N): the loads fly while the slower waves finish
      // phase 1, the images are overwritten after the barrier
    constexpr int RPP = NT / 8;
    const int c8 = tid & 7, r_in = tid >> 3;
    typename P::Raw va[NKP], vb[NKP];
#pragma unroll
    for (int p = 0; p < NKP; ++p) {
      const int qr = p * RPP + r_in;
      va[p] = P::zero_raw();
      vb[p] = va[p];
      if (qr <= N) {
        const int tk = qr < N ? tok0 + qr : 0;
        va[p] = P::load_raw(base + (size_t)tk * ts + c8 * 8);
        vb[p] = P::load_raw(dobase + (size_t)tk * D + c8 * 8);
      }
    }
    __syncthreads();                                      // every wave is done with the K, V images
#pragma unroll
    for (int p = 0; p < NKP; ++p) {
      P::stage(img0, LO, img_off(p * RPP + r_in, c8), va[p]);
      P::stage(img1, LO, img_off(p * RPP + r_in, c8), vb[p]);
    }
  }
  if (dq_part && tid < 64)                               // the q scaling (head_dim^-0.5) rides here, as in the dQ stores
    dq_part[((size_t)b * F + f) * D + h * 64 + tid] =
        0.125f * ((qsum_s[tid] + qsum_s[64 + tid]) + (qsum_s[128 + tid] + qsum_s[192 + tid]));
  __syncthreads();

  const int cls_qp = N >> 5, cls_sub = N & 31;            // where the cls query sits in the pair loop
#pragma unroll 1
  for (int kp = wave; kp < NKP; kp += NW) {
    if (kp != wave) load_kv(kp);
    f32x4 adk[2][4], adv[2][4];                          // dK^T, dV^T: [channel dt*16 + g*4 + r][key c]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { adk[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; adv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int qp = 0; qp < nqp; ++qp) {
      const uint16_t* Qp = img0 + qp * 32 * RS;          // 32-query slab of the two images
      const uint16_t* Gp = img1 + qp * 32 * RS;
      Op qa[2][2], ga[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          qa[qt][hh] = P::tile_op(Qp, LO, qt, fo.a[hh]);
          ga[qt][hh] = P::tile_op(Gp, LO, qt, fo.a[hh]);
        }
      const float4 ls0 = *reinterpret_cast<const float4*>(lse_s + qp * 32 + g * 4);
      const float4 ls1 = *reinterpret_cast<const float4*>(lse_s + qp * 32 + 16 + g * 4);
      const float4 de0 = *reinterpret_cast<const float4*>(del_s + qp * 32 + g * 4);
      const float4 de1 = *reinterpret_cast<const float4*>(del_s + qp * 32 + 16 + g * 4);
      const float lsa[8] = {ls0.x, ls0.y, ls0.z, ls0.w, ls1.x, ls1.y, ls1.z, ls1.w};
      const float dea[8] = {de0.x, de0.y, de0.z, de0.w, de1.x, de1.y, de1.z, de1.w};
      // (cls query, cls key) outside frame 0 is not attended: one element of key tile 0
      const bool kill_pair = f != 0 && kp == 0 && qp == cls_qp && c == 0;
      Op pa[2], da[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        f32x4 p0 = {-dea[0], -dea[1], -dea[2], -dea[3]}, p1 = {-dea[4], -dea[5], -dea[6], -dea[7]};    // dP - delta
        s0 = mfma(qa[0][0], kk[t][0], s0);
        s1 = mfma(qa[1][0], kk[t][0], s1);
        p0 = mfma(ga[0][0], vv[t][0], p0);
        p1 = mfma(ga[1][0], vv[t][0], p1);
        s0 = mfma(qa[0][1], kk[t][1], s0);
        s1 = mfma(qa[1][1], kk[t][1], s1);
        p0 = mfma(ga[0][1], vv[t][1], p0);
        p1 = mfma(ga[1][1], vv[t][1], p1);
        float e0[4], e1[4], d0[4], d1[4];
#if SBW_PACKED
        const f32x2 k2 = {kExp2, kExp2};
#endif
#pragma unroll
        for (int r = 0; r < 4; r += 2) {          // score pairs
#if SBW_PACKED
          const f32x2 a0 = f32x2{s0[r], s0[r + 1]} * k2 - f32x2{lsa[r], lsa[r + 1]};
          const f32x2 a1 = f32x2{s1[r], s1[r + 1]} * k2 - f32x2{lsa[4 + r], lsa[5 + r]};
#else
          const float a0[2] = {fmaf(s0[r], kExp2, -lsa[r]), fmaf(s0[r + 1], kExp2, -lsa[r + 1])};
          const float a1[2] = {fmaf(s1[r], kExp2, -lsa[4 + r]), fmaf(s1[r + 1], kExp2, -lsa[5 + r])};
#endif
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            e0[r + u] = __builtin_amdgcn_exp2f(a0[u]);
            e1[r + u] = __builtin_amdgcn_exp2f(a1[u]);
            if (t == 0) {
              e0[r + u] = (kill_pair && g * 4 + r + u == cls_sub) ? 0.f : e0[r + u];
              e1[r + u] = (kill_pair && 16 + g * 4 + r + u == cls_sub) ? 0.f : e1[r + u];
            }
          }
#if SBW_PACKED
          const f32x2 x0 = f32x2{e0[r], e0[r + 1]} * f32x2{p0[r], p0[r + 1]};
          const f32x2 x1 = f32x2{e1[r], e1[r + 1]} * f32x2{p1[r], p1[r + 1]};
          d0[r] = x0[0]; d0[r + 1] = x0[1];
          d1[r] = x1[0]; d1[r + 1] = x1[1];
#else
          d0[r] = e0[r] * p0[r]; d0[r + 1] = e0[r + 1] * p0[r + 1];
          d1[r] = e1[r] * p1[r]; d1[r + 1] = e1[r + 1] * p1[r + 1];
#endif
        }
        pa[t] = P::pack(e0, e1);
        da[t] = P::pack(d0, d1);
      }
      // dV^T += dO^T P, dK^T += Q^T dS: every transpose read feeds both key tiles
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const Tr g_lo = P::tile_tr(Gp, LO, 0, fo.tr[dt]), g_hi = P::tile_tr(Gp, LO, 1, fo.tr[dt]);
        const Tr q_lo = P::tile_tr(Qp, LO, 0, fo.tr[dt]), q_hi = P::tile_tr(Qp, LO, 1, fo.tr[dt]);
        const Op gb = P::join(g_lo, g_hi), qb = P::join(q_lo, q_hi);
        adv[0][dt] = mfma(gb, pa[0], adv[0][dt]);
        adk[0][dt] = mfma(qb, da[0], adk[0][dt]);
        adv[1][dt] = mfma(gb, pa[1], adv[1][dt]);
        adk[1][dt] = mfma(qb, da[1], adk[1][dt]);
      }
    }

    io_t* dkb = dqkv + (size_t)b * T * ts + D + h * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int krow = (2 * kp + t) * 16 + c;
      if (krow >= 1 && krow < nkeys) {
        io_t* row = dkb + (size_t)(tok0 + krow - 1) * ts;
        store_token_channels<P>(row, adk[t], 0.125f, g);
        store_token_channels<P>(row + D, adv[t], 1.0f, g);
      } else if (krow == 0) {          // the cls KEY collects gradient from every frame: this frame's slot
        slab_token_channels(cls_ws + 64, adk[t], 0.125f, g);
        slab_token_channels(cls_ws + 128, adv[t], 1.0f, g);
      }
    }
  }
}

template <typename P, int NKP>
int launch_fused(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* atom_ws,
                 float* dq_part, int B, int F, int N, int H, hipStream_t st) {
  using L = FusedLds<NKP, P::kImages>;
  using io_t = typename P::io_t;
  static_assert(L::total <= 160 * 1024, "LDS per CU");
  if (L::total > 64 * 1024)
    if (int rc = lvl_allow_lds<space_bwd_fused_kernel<P, NKP>>()) return rc;
  hipLaunchKernelGGL((space_bwd_fused_kernel<P, NKP>), dim3((unsigned)(B * F * H)), dim3(256), L::total, st,
                     (const io_t*)qkv, (const io_t*)out, (const io_t*)dout, lse, (io_t*)dqkv, atom_ws, dq_part, F, N, H);
  LVL_CHECK_LAUNCH("space_bwd_fused");
  return LVL_OK;
}

constexpr int kFusedPairs = 9;         // fused kernel: up to 288 keys per group (bf16 and f32-class alike)

template <typename P>
int dispatch_fused(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* atom_ws,
                   float* dq_part, int B, int F, int N, int H, hipStream_t st) {
  switch ((N + 1 + 31) / 32) {
#define SPACE_FUSED_CASE(K) case K: return launch_fused<P, K>(qkv, out, dout, lse, dqkv, atom_ws, dq_part, B, F, N, H, st);
    SPACE_FUSED_CASE(1) SPACE_FUSED_CASE(2) SPACE_FUSED_CASE(3) SPACE_FUSED_CASE(4) SPACE_FUSED_CASE(5)
    SPACE_FUSED_CASE(6) SPACE_FUSED_CASE(7) SPACE_FUSED_CASE(8) SPACE_FUSED_CASE(9)
#undef SPACE_FUSED_CASE
  }
  return lvl_fail(LVL_ENOSYS, "space_mfma_bwd: %d keys per group not supported by the fused kernel", N + 1);
}

// dqkv[b, token 0, :] = (d cls q | d cls k | d cls v) from the f32 atomic workspace
template <typename P>
__global__ __launch_bounds__(192) void cls_grad_finalize_kernel(const float* __restrict__ atom_ws,
                                                                typename P::io_t* __restrict__ dqkv, int T, int H,
                                                                int nslots) {
  // The cls token's d(q | k | v) of one (b, h): the sum of `nslots` partial records [192] f32, one per contributing
  // workgroup (space kernels: one per frame; time kernels: one per location chunk), added up in slot order. Round 6: the
  // contributors used f32 atomicAdd on ONE record -- the order of four to twenty-five additions then depended on timing, and
  // through the cls query of the time attention a last-bit difference flipped a bf16 rounding of dqkv about one step in six
  // at the TSF-B geometry: the "second outcome" of the step (profiles/r06_second_outcome.txt).
  const int h = blockIdx.x % H, b = blockIdx.x / H, t = threadIdx.x;      // t in [0,192): part = t/64
  const int D = H * 64;
  const float* rec = atom_ws + ((size_t)b * H + h) * nslots * 192 + t;
  float acc = 0.f;
  for (int s = 0; s < nslots; ++s) acc += rec[(size_t)s * 192];
  dqkv[(size_t)b * T * 3 * D + (t >> 6) * D + h * 64 + (t & 63)] = P::from_f32(acc);
}

template <typename P, int NKT, bool TEXT = false, int NW = 8, bool MASKALL = false>
int launch_dq(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta, int B,
              int F, int N, int H, hipStream_t st) {
  using L = DqLds<NKT, NW, P::kImages>;
  using io_t = typename P::io_t;
  static_assert(L::total <= 160 * 1024, "LDS per CU");
  if (L::total > 64 * 1024)
    if (int rc = lvl_allow_lds<space_bwd_dq_kernel<P, NKT, TEXT, NW, MASKALL>>()) return rc;
  hipLaunchKernelGGL((space_bwd_dq_kernel<P, NKT, TEXT, NW, MASKALL>), dim3((unsigned)(B * F * H)), dim3(NW * 64),
                     L::total, st, (const io_t*)qkv, (const io_t*)out, (const io_t*)dout, lse, (io_t*)dqkv, delta, F, N,
                     H);
  LVL_CHECK_LAUNCH("space_bwd_dq");
  return LVL_OK;
}

constexpr int kBigTiles = 37;          // large-group variant: up to 592 keys, 4 waves, one workgroup per CU

template <typename P, bool TEXT>
int launch_dkv(const void* qkv, const void* out, const void* dout, const float* lse, const float* delta, void* dqkv,
               float* atom_ws, int B, int F, int N, int H, hipStream_t st) {
  using io_t = typename P::io_t;
  const bool big = !TEXT && N + 1 > 272;
  const DkvGeom G = dkv_geometry(N, big ? 4 : 8, P::kImages);
  if (big) {
    if (int rc = lvl_allow_lds<space_bwd_dkv_kernel<P, TEXT, 4>>()) return rc;
    hipLaunchKernelGGL((space_bwd_dkv_kernel<P, TEXT, 4>), dim3((unsigned)(B * F * H)), dim3(256), G.total, st,
                       (const io_t*)qkv, (const io_t*)out, (const io_t*)dout, lse, delta, (io_t*)dqkv, atom_ws, F, N, H,
                       G);
  } else {
    if (G.total > 64 * 1024)
      if (int rc = lvl_allow_lds<space_bwd_dkv_kernel<P, TEXT, 8>>()) return rc;
    hipLaunchKernelGGL((space_bwd_dkv_kernel<P, TEXT, 8>), dim3((unsigned)(B * F * H)), dim3(512), G.total, st,
                       (const io_t*)qkv, (const io_t*)out, (const io_t*)dout, lse, delta, (io_t*)dqkv, atom_ws, F, N, H,
                       G);
  }
  LVL_CHECK_LAUNCH("space_bwd_dkv");
  return LVL_OK;
}

template <typename P, bool TEXT>
int dispatch_dq(int nkeys, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                float* delta, int B, int F, int N, int H, hipStream_t st) {
  if constexpr (TEXT) switch ((nkeys + 15) / 16) {      // exact tile count: the kernel masks only the last key tile
#define SPACE_DQ_CASE(K) case K: return launch_dq<P, K, TEXT>(qkv, out, dout, lse, dqkv, delta, B, F, N, H, st);
    SPACE_DQ_CASE(1) SPACE_DQ_CASE(2) SPACE_DQ_CASE(3) SPACE_DQ_CASE(4) SPACE_DQ_CASE(5) SPACE_DQ_CASE(6) SPACE_DQ_CASE(7)
    SPACE_DQ_CASE(8) SPACE_DQ_CASE(9) SPACE_DQ_CASE(10) SPACE_DQ_CASE(11) SPACE_DQ_CASE(12) SPACE_DQ_CASE(13)
    SPACE_DQ_CASE(14) SPACE_DQ_CASE(15) SPACE_DQ_CASE(16) SPACE_DQ_CASE(17)
#undef SPACE_DQ_CASE
  }
  if constexpr (!TEXT && !P::kSplit) {
    if ((nkeys + 15) / 16 == kBigTiles)
      return launch_dq<P, kBigTiles, false, 4, false>(qkv, out, dout, lse, dqkv, delta, B, F, N, H, st);
    if (nkeys <= kBigTiles * 16)
      return launch_dq<P, kBigTiles, false, 4, true>(qkv, out, dout, lse, dqkv, delta, B, F, N, H, st);
  }
  return lvl_fail(LVL_ENOSYS, "space_mfma_bwd: %d keys per group not supported", nkeys);
}

}  // namespace

void lvl_launch_cls_grad_finalize(const float* atom_ws, void* dqkv, int B, int T, int H, int nslots, int dtype,
                                  hipStream_t st) {
  if (dtype == LVL_F32)
    hipLaunchKernelGGL(cls_grad_finalize_kernel<PrecSplit>, dim3((unsigned)(B * H)), dim3(192), 0, st, atom_ws,
                       (float*)dqkv, T, H, nslots);
  else
    hipLaunchKernelGGL(cls_grad_finalize_kernel<PrecBf16>, dim3((unsigned)(B * H)), dim3(192), 0, st, atom_ws,
                       (uint16_t*)dqkv, T, H, nslots);
}

// float32 (f32-class, PrecSplit): the fused kernel only (up to 288 keys; its four images fit the LDS)
bool lvl_space_mfma_bwd_supported(int F, int N, int dtype) {
  if (N < 1 || F > 64) return false;
  if (N + 1 <= kFusedPairs * 32) return true;                                      // fused kernel
  if (dtype == LVL_F32) return false;
  return N + 1 <= kBigTiles * 16 && dkv_geometry(N, 4).total <= 160 * 1024;      // large groups: 4-wave kernels
}

// rows of the dq column-sum slab the fused kernel writes (0: this shape runs on kernels without the rider)
int lvl_space_mfma_bwd_dq_part_rows(int B, int F, int N) { return N + 1 <= kFusedPairs * 32 ? B * F : 0; }

// ws layout: delta [B*H*T] f32, then the cls token's partial records [B*H][F][192] f32 (d cls q | d cls k | d cls v per frame)
// dq_part (nullable): [B*F, H*64] f32 partial column sums of dQ, written when lvl_space_mfma_bwd_dq_part_rows > 0
int lvl_space_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                       float* dq_part, int B, int F, int N, int H, int dtype, hipStream_t st) {
  const int T = 1 + F * N;
  float* delta = ws;
  float* atom_ws = ws + (size_t)B * H * T;            // every slot is written whole by its frame's workgroup: no zeroing
  if (N + 1 <= kFusedPairs * 32) {
    if (int rc = dtype == LVL_F32 ? dispatch_fused<PrecSplit>(qkv, out, dout, lse, dqkv, atom_ws, dq_part, B, F, N, H, st)
                                  : dispatch_fused<PrecBf16>(qkv, out, dout, lse, dqkv, atom_ws, dq_part, B, F, N, H, st))
      return rc;
  } else {
    if (dtype == LVL_F32) return lvl_fail(LVL_ENOSYS, "space_mfma_bwd (f32 class): %d keys per group", N + 1);
    if (int rc = dispatch_dq<PrecBf16, false>(N + 1, qkv, out, dout, lse, dqkv, delta, B, F, N, H, st)) return rc;
    if (int rc = launch_dkv<PrecBf16, false>(qkv, out, dout, lse, delta, dqkv, atom_ws, B, F, N, H, st)) return rc;
  }
  lvl_launch_cls_grad_finalize(atom_ws, dqkv, B, T, H, F, dtype, st);
  LVL_CHECK_LAUNCH("cls_grad_finalize");
  return LVL_OK;
}

bool lvl_text_mfma_bwd_supported(int L, int dtype) {
  return L >= 1 && L <= 256 && dkv_geometry(L, 8, dtype == LVL_F32 ? 2 : 1).total <= 160 * 1024;
}

// ws: delta [B*H*L] f32
int lvl_text_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int L, int H, int dtype, hipStream_t st) {
  if (dtype == LVL_F32) {
    if (int rc = dispatch_dq<PrecSplit, true>(L, qkv, out, dout, lse, dqkv, ws, B, 1, L, H, st)) return rc;
    return launch_dkv<PrecSplit, true>(qkv, out, dout, lse, ws, dqkv, nullptr, B, 1, L, H, st);
  }
  if (int rc = dispatch_dq<PrecBf16, true>(L, qkv, out, dout, lse, dqkv, ws, B, 1, L, H, st)) return rc;
  return launch_dkv<PrecBf16, true>(qkv, out, dout, lse, ws, dqkv, nullptr, B, 1, L, H, st);
}
