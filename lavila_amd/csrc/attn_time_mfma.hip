// Time-mode divided attention on the matrix cores for MANY frames (5 <= F <= 16; bf16 in/out, f32 accumulate), gfx950.
//
// Per (sample b, location n, head h): F queries x (1 cls + F) keys, head dim 64 (timesformer.py:121-131 with the
// '(b n) f d' grouping :302-303). The register-tiled kernels of attn_time.hip keep a whole problem in VALU
// registers, which is right up to 4 frames (2.5 flop/B) but VALU-bound from 8 frames on (F=16: 1.0-1.2 TB/s). Here
// one WAVE owns one problem at a time and the F x (F+1) products run as 16x16x32 MFMA tiles:
//   * fragments come straight from HBM: lane (c = lane & 15, g = lane >> 4) loads channels g*8..+7 and 32+g*8..+7 of
//     the q / k / v row of FRAME c -- exactly the A/B operand layout of K.Q^T -- so every 128-byte head row is read
//     once; rows of frames >= F are zero and masked (any F up to 16 runs on the same code);
//   * the cls key rides as key 16: its score is a VALU dot product, its probability sits in the upper half of the
//     P operand, and row 16 of the wave's LDS image of V (resp. K) holds the cls value (key) row, so the P.V (dS.K)
//     MFMAs add its term for free;
//   * contractions over keys / queries (P.V, dS.K, dS^T.Q, P^T.dO) read per-wave swizzled LDS images with the
//     transpose read (attn_mfma_common.h), the A operand being the freshly computed tile in its C layout;
//   * the CLS query (token 0, attends to every key of the sample) is one more query: its flash-style running partial
//     over the wave's locations is kept in registers (the 64 accumulated channels as row 0 of an MFMA tile) and
//     merged by cls_combine_kernel; in the backward its rank-1 terms ride as query 16 and d(cls q), d(cls k),
//     d(cls v) -- which collect gradient from every location -- are rows of extra MFMA tiles summed in registers and
//     added to the f32 workspace once per wave.
// A workgroup is 4 waves = 4 consecutive heads of the same location range (their head rows are adjacent in memory).
// HBM-bound: 4 rows in + 1 row out per token forward, 5 in + 3 out backward; the MFMA work is noise.
#include "attn_mfma_common.h"

namespace {

using namespace attn_mfma;
constexpr int CLS_REC = 66;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kExp2 = 0.125f * kLog2e;          // exp(s * scale) = exp2(s * kExp2)
constexpr int NWV = 4;                            // waves (= heads) per workgroup
constexpr int IMG = 32 * RS;                      // elements of one per-wave image: 32 rows x 64 channels

// max over each aligned group of 16 lanes (one DPP row); every lane of the group receives it
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  return v;
}
__device__ __forceinline__ float dot16(const uint4& a0, const uint4& a1, const float (&b)[16]) {
  float x[8], s = 0.f;
  Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&a0), x);
#pragma unroll
  for (int i = 0; i < 8; ++i) s = fmaf(x[i], b[i], s);
  Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&a1), x);
#pragma unroll
  for (int i = 0; i < 8; ++i) s = fmaf(x[i], b[8 + i], s);
  return s;
}
// the 16 channels of a head row this lane's fragments cover: g*8..+7 and 32+g*8..+7
__device__ __forceinline__ void load_slices(const uint16_t* row, int g, float (&v)[16]) {
  Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(row + g * 8), reinterpret_cast<float(&)[8]>(v[0]));
  Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(row + 32 + g * 8), reinterpret_cast<float(&)[8]>(v[8]));
}
// writes this lane's two fragments (channels g*8.. and 32+g*8..) of image row `row`
__device__ __forceinline__ void put_row(uint16_t* img, int row, int g, const uint4& f0, const uint4& f1) {
  *reinterpret_cast<uint4*>(img + img_off(row, g)) = f0;
  *reinterpret_cast<uint4*>(img + img_off(row, 4 + g)) = f1;
}
__device__ __forceinline__ uint4 tr_pair(const uint16_t* img, int off) {      // rows 0..15 | rows 16..31 of an image
  const uint2 lo = tile_frag_tr(img, 0, off), hi = tile_frag_tr(img, 1, off);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NWV * 64) void time_mfma_fwd_kernel(const uint16_t* __restrict__ qkv,
                                                                 uint16_t* __restrict__ out, float* __restrict__ lse,
                                                                 float* __restrict__ cls_ws, int F, int N, int H,
                                                                 int NCH, int NC) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[NWV * (IMG + 16 * OS)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int hb = blockIdx.x % (H / NWV), chunk = (blockIdx.x / (H / NWV)) % NC, b = blockIdx.x / ((H / NWV) * NC);
  const int h = hb * NWV + wave;
  const int D = H * 64, T = 1 + F * N;
  const size_t ts = (size_t)3 * D;
  const uint16_t* base = qkv + (size_t)b * T * ts + h * 64;
  uint16_t* Vi = smem + wave * (IMG + 16 * OS);
  uint16_t* ot = Vi + IMG;
  const FragOff fo = frag_offsets(lane);
  const bool live = c < F;                        // this lane's frame exists

  // cls key / query slices (the 16 channels this lane's fragments cover), cls value -> image row 16, rows 17..31 = 0
  float kc[16], qc[16];
  load_slices(base + D, g, kc);
  load_slices(base, g, qc);
  {
    const uint4 z = make_uint4(0, 0, 0, 0);
    const uint16_t* vrow = base + 2 * D;
    put_row(Vi, 16 + c, g, c == 0 ? *reinterpret_cast<const uint4*>(vrow + g * 8) : z,
            c == 0 ? *reinterpret_cast<const uint4*>(vrow + 32 + g * 8) : z);
  }
  // running partial of the CLS query over this wave's keys: max (raw score units), sum, acc = row 0 of an MFMA tile
  float cm = -INFINITY, cl = 0.f;
  f32x4 ca[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) ca[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (chunk == 0) {                               // the cls key itself enters the CLS row exactly once per (b, h)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = fmaf(qc[i], kc[i], s);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    cm = s;
    cl = 1.f;
    if (g == 0) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ca[dt][0] = bf16_to_f32(base[2 * D + dt * 16 + c]);
    }
  }

  const int n_begin = chunk * NCH, n_end = min(N, (chunk + 1) * NCH);
  uint4 nq0, nq1, nk0, nk1, nv0, nv1;
  auto load_problem = [&](int n) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    nq0 = nq1 = nk0 = nk1 = nv0 = nv1 = z;
    if (live) {
      const uint16_t* p = base + (size_t)(1 + c * N + n) * ts + g * 8;
      nq0 = *reinterpret_cast<const uint4*>(p); nq1 = *reinterpret_cast<const uint4*>(p + 32);
      nk0 = *reinterpret_cast<const uint4*>(p + D); nk1 = *reinterpret_cast<const uint4*>(p + D + 32);
      nv0 = *reinterpret_cast<const uint4*>(p + 2 * D); nv1 = *reinterpret_cast<const uint4*>(p + 2 * D + 32);
    }
  };
  if (n_begin < n_end) load_problem(n_begin);
#pragma unroll 1
  for (int n = n_begin; n < n_end; ++n) {
    const uint4 q0 = nq0, q1 = nq1, k0 = nk0, k1 = nk1, v0 = nv0, v1 = nv1;
    if (n + 1 < n_end) load_problem(n + 1);       // next location's rows are in flight under this one's work
    put_row(Vi, c, g, v0, v1);                    // V rows -> image (same wave reads it back: LDS ops are ordered)

    // S^T = K.Q^T: st[r] = raw score of query c against key g*4+r; cls key: sc (all four lanes of query c)
    f32x4 st = mfma(k0, q0, f32x4{0.f, 0.f, 0.f, 0.f});
    st = mfma(k1, q1, st);
    float sc = dot16(q0, q1, kc);
    sc += __shfl_xor(sc, 16, 64);
    sc += __shfl_xor(sc, 32, 64);
    float m = sc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (g * 4 + r >= F) st[r] = -INFINITY;      // keys of frames that do not exist
      m = fmaxf(m, st[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mk = m * kExp2;
    float p[4], l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(fmaf(st[r], kExp2, -mk)); l += p[r]; }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float pc = __builtin_amdgcn_exp2f(fmaf(sc, kExp2, -mk));
    l += pc;
    // O = P.V: keys 0..15 in the lower half of the operand, the cls key as key 16 (lane group g == 0, first slot)
    const uint4 pa = make_uint4(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), g == 0 ? pack_bf16x2(pc, 0.f) : 0u, 0u);

    // CLS query over this location's F keys: scores live on the key lanes (c = key)
    float s2 = dot16(k0, k1, qc);
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (!live) s2 = -INFINITY;
    const float mn = fmaxf(cm, row16_max(s2));
    const float al = (cm == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((cm - mn) * kExp2);
    const float p2 = __builtin_amdgcn_exp2f((s2 - mn) * kExp2);        // probability of key c (0 for dead frames)
    cl = cl * al + row16_sum(p2);
    cm = mn;
    // its P operand: query column 0 only; lane (c = 0, g) needs the probabilities of keys g*4 .. g*4+3
    const float e0 = __shfl(p2, g * 4 + 0, 64), e1 = __shfl(p2, g * 4 + 1, 64), e2 = __shfl(p2, g * 4 + 2, 64),
                e3 = __shfl(p2, g * 4 + 3, 64);
    const uint4 pa2 = c == 0 ? make_uint4(pack_bf16x2(e0, e1), pack_bf16x2(e2, e3), 0u, 0u) : make_uint4(0, 0, 0, 0);

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const uint4 vt = tr_pair(Vi, fo.tr[dt]);
      o[dt] = mfma(pa, vt, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 0; r < 4; ++r) ca[dt][r] *= al;
      ca[dt] = mfma(pa2, vt, ca[dt]);
    }
    // normalise, transpose through LDS, store whole rows (row = frame)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float linv = __builtin_amdgcn_rcpf(__shfl(l, g * 4 + r, 64));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt][r] *= linv;
    }
    store_tile_rows<PrecBf16>(ot, o, 1.0f, lane,
                    [&](int row) { return out + ((size_t)b * T + 1 + (size_t)row * N + n) * D + h * 64; },
                    [&](int row) { return row < F; });
    if (g == 0 && live) lse[((size_t)b * H + h) * T + 1 + c * N + n] = m * 0.125f + __logf(l);
  }
  // partial record of the CLS query over this wave's locations
  float* rec = cls_ws + (((size_t)b * H + h) * NC + chunk) * CLS_REC;
  if (g == 0) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) rec[2 + dt * 16 + c] = ca[dt][0];
    if (c == 0) { rec[0] = cm * 0.125f; rec[1] = cl; }
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------
// Per problem, with P recomputed from the saved row log-sum-exp:
//   orientation A (query per lane):  S^T = K.Q^T, dP^T = V.dO^T  ->  delta, dS^T  ->  dQ = dS.K      (+ cls key 16)
//   orientation B (key per lane):    S = Q.K^T,   dP = dO.V^T    ->  P, dS       ->  dV = P^T.dO, dK = dS^T.Q
//                                                                                     (+ cls query as query 16)
//   cls rows: d(cls k) += dS_c^T.Q, d(cls v) += P_c^T.dO (row 0 of tiles whose A operand is the cls-key column),
//             d(cls q) += dS_cls.K (row 0 of a tile whose A operand is the cls-query row).
__global__ __launch_bounds__(NWV * 64) void time_mfma_bwd_kernel(const uint16_t* __restrict__ qkv,
                                                                 const uint16_t* __restrict__ out,
                                                                 const uint16_t* __restrict__ dout,
                                                                 const float* __restrict__ lse,
                                                                 uint16_t* __restrict__ dqkv, float* __restrict__ atom_ws,
                                                                 int F, int N, int H, int NCH, int NC) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];      // NWV x (4 images + transposition tile)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int hb = blockIdx.x % (H / NWV), chunk = (blockIdx.x / (H / NWV)) % NC, b = blockIdx.x / ((H / NWV) * NC);
  const int h = hb * NWV + wave;
  const int D = H * 64, T = 1 + F * N;
  const size_t ts = (size_t)3 * D;
  const uint16_t* base = qkv + (size_t)b * T * ts + h * 64;
  const uint16_t* dobase = dout + (size_t)b * T * D + h * 64;
  uint16_t* gbase = dqkv + (size_t)b * T * ts + h * 64;
  const float* lrow = lse + ((size_t)b * H + h) * T;
  uint16_t* Qi = smem + wave * (4 * IMG + 16 * OS);
  uint16_t* Ki = Qi + IMG;
  uint16_t* Vi = Ki + IMG;
  uint16_t* Gi = Vi + IMG;                        // dO image
  uint16_t* ot = Gi + IMG;
  const FragOff fo = frag_offsets(lane);
  const bool live = c < F;

  // cls rows of this head: slices for the VALU dots; image rows 16 (q, k, v, dO of the cls token), rows 17..31 = 0
  float kc[16], qc[16], vc[16], gc[16];
  load_slices(base, g, qc);
  load_slices(base + D, g, kc);
  load_slices(base + 2 * D, g, vc);
  load_slices(dobase, g, gc);
  {
    const uint4 z = make_uint4(0, 0, 0, 0);
    auto put16 = [&](uint16_t* img, const uint16_t* row) {
      put_row(img, 16 + c, g, c == 0 ? *reinterpret_cast<const uint4*>(row + g * 8) : z,
              c == 0 ? *reinterpret_cast<const uint4*>(row + 32 + g * 8) : z);
    };
    put16(Qi, base);
    put16(Ki, base + D);
    put16(Vi, base + 2 * D);
    put16(Gi, dobase);
  }
  float dlc;                                     // delta of the cls row = dO_cls . O_cls
  {
    float oc[16], t = 0.f;
    load_slices(out + (size_t)b * T * D + h * 64, g, oc);
#pragma unroll
    for (int i = 0; i < 16; ++i) t = fmaf(gc[i], oc[i], t);
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    dlc = t;
  }
  const float Lc2 = lrow[0] * kLog2e;            // cls-row lse in log2 units

  // accumulators of the cls token's gradients (row 0 of the tiles): d(cls q), d(cls k), d(cls v)
  f32x4 aq[4], ak[4], av[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { aq[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; ak[dt] = aq[dt]; av[dt] = aq[dt]; }
  if (chunk == 0 && g == 0) {                    // the cls key inside the CLS row, once per (b, h)
    // p = exp(q_c.k_c*scale - lse_c), ds = p*(dO_c.v_c - delta_c): d cls v += p dO_c, d cls k += ds q_c, d cls q += ds k_c
    float s = 0.f, dp = 0.f;
    for (int i = 0; i < 64; ++i) {
      s = fmaf(bf16_to_f32(base[i]), bf16_to_f32(base[D + i]), s);
      dp = fmaf(bf16_to_f32(dobase[i]), bf16_to_f32(base[2 * D + i]), dp);
    }
    const float p = __builtin_amdgcn_exp2f(fmaf(s, kExp2, -Lc2)), ds = p * (dp - dlc);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      av[dt][0] = p * bf16_to_f32(dobase[dt * 16 + c]);
      ak[dt][0] = ds * bf16_to_f32(base[dt * 16 + c]);
      aq[dt][0] = ds * bf16_to_f32(base[D + dt * 16 + c]);
    }
  }

  const int n_begin = chunk * NCH, n_end = min(N, (chunk + 1) * NCH);
  uint4 nq0, nq1, nk0, nk1, nv0, nv1, ng0, ng1;
  float nl;
  auto load_problem = [&](int n) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    nq0 = nq1 = nk0 = nk1 = nv0 = nv1 = ng0 = ng1 = z;
    nl = INFINITY;                                // dead frames: exp2(s - inf) = 0
    if (live) {
      const int tok = 1 + c * N + n;
      const uint16_t* p = base + (size_t)tok * ts + g * 8;
      nq0 = *reinterpret_cast<const uint4*>(p); nq1 = *reinterpret_cast<const uint4*>(p + 32);
      nk0 = *reinterpret_cast<const uint4*>(p + D); nk1 = *reinterpret_cast<const uint4*>(p + D + 32);
      nv0 = *reinterpret_cast<const uint4*>(p + 2 * D); nv1 = *reinterpret_cast<const uint4*>(p + 2 * D + 32);
      const uint16_t* gp = dobase + (size_t)tok * D + g * 8;
      ng0 = *reinterpret_cast<const uint4*>(gp); ng1 = *reinterpret_cast<const uint4*>(gp + 32);
      nl = lrow[tok] * kLog2e;
    }
  };
  if (n_begin < n_end) load_problem(n_begin);
#pragma unroll 1
  for (int n = n_begin; n < n_end; ++n) {
    const uint4 q0 = nq0, q1 = nq1, k0 = nk0, k1 = nk1, v0 = nv0, v1 = nv1, g0 = ng0, g1 = ng1;
    const float Lq = nl;                          // lse (log2 units) of query c
    if (n + 1 < n_end) load_problem(n + 1);
    put_row(Qi, c, g, q0, q1);
    put_row(Ki, c, g, k0, k1);
    put_row(Vi, c, g, v0, v1);
    put_row(Gi, c, g, g0, g1);

    // ---- orientation A: lane (c = query, g), rows = keys g*4+r ---------------------------------------------------
    f32x4 st = mfma(k0, q0, f32x4{0.f, 0.f, 0.f, 0.f});
    st = mfma(k1, q1, st);
    f32x4 dpt = mfma(v0, g0, f32x4{0.f, 0.f, 0.f, 0.f});
    dpt = mfma(v1, g1, dpt);
    float sc = dot16(q0, q1, kc), dpc = dot16(g0, g1, vc);
    sc += __shfl_xor(sc, 16, 64); sc += __shfl_xor(sc, 32, 64);
    dpc += __shfl_xor(dpc, 16, 64); dpc += __shfl_xor(dpc, 32, 64);
    float pt[4], dl = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pt[r] = (g * 4 + r < F) ? __builtin_amdgcn_exp2f(fmaf(st[r], kExp2, -Lq)) : 0.f;
      dl = fmaf(pt[r], dpt[r], dl);
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float pcq = __builtin_amdgcn_exp2f(fmaf(sc, kExp2, -Lq));    // P[query c][cls key] (0 for dead queries)
    dl = fmaf(pcq, dpc, dl);                       // delta_q = sum_j P_qj dP_qj over all F+1 keys
    const float dscq = pcq * (dpc - dl);           // dS[query c][cls key]
    float dst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = pt[r] * (dpt[r] - dl);
    // dQ = dS.K (keys 0..15 | cls key as key 16, K image row 16 = cls key)
    const uint4 pa_ds = make_uint4(pack_bf16x2(dst[0], dst[1]), pack_bf16x2(dst[2], dst[3]),
                                   g == 0 ? pack_bf16x2(dscq, 0.f) : 0u, 0u);
    // d(cls k) += sum_q dS[q][cls] Q[q], d(cls v) += sum_q P[q][cls] dO[q]: A operand = the cls-key column as row 0
    // (lane (c = 0, g) carries queries g*4 .. g*4+3)
    const float x0 = __shfl(dscq, g * 4 + 0, 64), x1 = __shfl(dscq, g * 4 + 1, 64), x2 = __shfl(dscq, g * 4 + 2, 64),
                x3 = __shfl(dscq, g * 4 + 3, 64);
    const float y0 = __shfl(pcq, g * 4 + 0, 64), y1 = __shfl(pcq, g * 4 + 1, 64), y2 = __shfl(pcq, g * 4 + 2, 64),
                y3 = __shfl(pcq, g * 4 + 3, 64);
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    const uint4 pa_ck = c == 0 ? make_uint4(pack_bf16x2(x0, x1), pack_bf16x2(x2, x3), 0u, 0u) : z4;
    const uint4 pa_cv = c == 0 ? make_uint4(pack_bf16x2(y0, y1), pack_bf16x2(y2, y3), 0u, 0u) : z4;

    // ---- orientation B: lane (c = key, g), rows = queries g*4+r; the cls query rides as query 16 ----------------------
    f32x4 sb = mfma(q0, k0, f32x4{0.f, 0.f, 0.f, 0.f});
    sb = mfma(q1, k1, sb);
    f32x4 dpb = mfma(g0, v0, f32x4{0.f, 0.f, 0.f, 0.f});
    dpb = mfma(g1, v1, dpb);
    float pb[4], dsb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float Lr = __shfl(Lq, g * 4 + r, 64), dr = __shfl(dl, g * 4 + r, 64);   // lse / delta of query g*4+r
      pb[r] = __builtin_amdgcn_exp2f(fmaf(sb[r], kExp2, -Lr));       // dead queries: Lr = inf -> 0
      dsb[r] = pb[r] * (dpb[r] - dr);
    }
    // cls query against key c
    float s2 = dot16(k0, k1, qc), dp2 = dot16(v0, v1, gc);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    dp2 += __shfl_xor(dp2, 16, 64); dp2 += __shfl_xor(dp2, 32, 64);
    const float p2 = live ? __builtin_amdgcn_exp2f(fmaf(s2, kExp2, -Lc2)) : 0.f;
    const float ds2 = p2 * (dp2 - dlc);
    const uint4 pa_p = make_uint4(pack_bf16x2(pb[0], pb[1]), pack_bf16x2(pb[2], pb[3]), g == 0 ? pack_bf16x2(p2, 0.f) : 0u, 0u);
    const uint4 pa_d = make_uint4(pack_bf16x2(dsb[0], dsb[1]), pack_bf16x2(dsb[2], dsb[3]),
                                  g == 0 ? pack_bf16x2(ds2, 0.f) : 0u, 0u);
    // d(cls q) += sum_k dS_cls[k] K[k]: A operand = the cls-query row as row 0 (lane (c = 0, g): keys g*4 .. g*4+3)
    const float w0 = __shfl(ds2, g * 4 + 0, 64), w1 = __shfl(ds2, g * 4 + 1, 64), w2 = __shfl(ds2, g * 4 + 2, 64),
                w3 = __shfl(ds2, g * 4 + 3, 64);
    const uint4 pa_cq = c == 0 ? make_uint4(pack_bf16x2(w0, w1), pack_bf16x2(w2, w3), 0u, 0u) : z4;

    f32x4 odq[4], odk[4], odv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const uint4 kt = tr_pair(Ki, fo.tr[dt]), qt = tr_pair(Qi, fo.tr[dt]), gt = tr_pair(Gi, fo.tr[dt]);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      odq[dt] = mfma(pa_ds, kt, zero);
      odk[dt] = mfma(pa_d, qt, zero);
      odv[dt] = mfma(pa_p, gt, zero);
      aq[dt] = mfma(pa_cq, kt, aq[dt]);
      ak[dt] = mfma(pa_ck, qt, ak[dt]);
      av[dt] = mfma(pa_cv, gt, av[dt]);
    }
    auto row_ptr = [&](int row, int third) { return gbase + (size_t)(1 + (size_t)row * N + n) * ts + third * D; };
    store_tile_rows<PrecBf16>(ot, odq, 0.125f, lane, [&](int row) { return row_ptr(row, 0); }, [&](int row) { return row < F; });
    store_tile_rows<PrecBf16>(ot, odk, 0.125f, lane, [&](int row) { return row_ptr(row, 1); }, [&](int row) { return row < F; });
    store_tile_rows<PrecBf16>(ot, odv, 1.0f, lane, [&](int row) { return row_ptr(row, 2); }, [&](int row) { return row < F; });
  }
  // the cls token's gradients: row 0 of the accumulated tiles (lanes g == 0) -> this chunk's slot of the (b, h) partial
  // slab, one writer per slot; cls_grad_finalize_kernel adds the NC slots up in order (round 6: no f32 atomics)
  if (g == 0) {
    float* dst = atom_ws + (((size_t)b * H + h) * NC + chunk) * 192;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dst[dt * 16 + c] = aq[dt][0] * 0.125f;
      dst[64 + dt * 16 + c] = ak[dt][0] * 0.125f;
      dst[128 + dt * 16 + c] = av[dt][0];
    }
  }
}

struct Geo { int NCH, NC; };
inline Geo geometry(int N) {
  int nch = 16;                                // locations per wave: amortises the cls prologue / epilogue
  while ((N + nch - 1) / nch > 64) nch *= 2;   // at most 64 partial records per (b, h)
  return Geo{nch, (N + nch - 1) / nch};
}

}  // namespace

void lvl_launch_cls_combine(const float* ws, void* out, float* lse, int B, int H, int nparts, int T, int dtype,
                            hipStream_t st);
void lvl_launch_cls_grad_finalize(const float* atom_ws, void* dqkv, int B, int T, int H, int nslots, int dtype,
                                  hipStream_t st);

bool lvl_time_mfma_supported(int F, int N, int H) { return F >= 5 && F <= 16 && N >= 1 && H % NWV == 0; }

int lvl_time_mfma_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st) {
  const Geo g = geometry(N);
  hipLaunchKernelGGL(time_mfma_fwd_kernel, dim3((unsigned)(B * g.NC * (H / NWV))), dim3(NWV * 64), 0, st,
                     (const uint16_t*)qkv, (uint16_t*)out, lse, ws, F, N, H, g.NCH, g.NC);
  LVL_CHECK_LAUNCH("time_mfma_fwd");
  lvl_launch_cls_combine(ws, out, lse, B, H, g.NC, 1 + F * N, LVL_BF16, st);
  LVL_CHECK_LAUNCH("cls_combine");
  return LVL_OK;
}

// ws layout: delta [B*H*T] f32 (unused here), then the cls token's partial records [B*H][NC][192] f32
int lvl_time_mfma_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                      int B, int F, int N, int H, hipStream_t st) {
  const Geo g = geometry(N);
  const int T = 1 + F * N;
  float* atom_ws = ws + (size_t)B * H * T;            // every (b, h, chunk) slot is written whole by its wave: no zeroing
  if (int rc = lvl_allow_lds<time_mfma_bwd_kernel>()) return rc;
  hipLaunchKernelGGL(time_mfma_bwd_kernel, dim3((unsigned)(B * g.NC * (H / NWV))), dim3(NWV * 64),
                     NWV * (4 * IMG + 16 * OS) * sizeof(uint16_t), st,
                     (const uint16_t*)qkv, (const uint16_t*)out, (const uint16_t*)dout, lse, (uint16_t*)dqkv, atom_ws, F,
                     N, H, g.NCH, g.NC);
  LVL_CHECK_LAUNCH("time_mfma_bwd");
  lvl_launch_cls_grad_finalize(atom_ws, dqkv, B, T, H, g.NC, LVL_BF16, st);
  LVL_CHECK_LAUNCH("cls_grad_finalize");
  return LVL_OK;
}
