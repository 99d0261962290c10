// Space-mode divided attention forward on the matrix cores (bf16 in, f32 accumulate), gfx950.
//
// One 512-thread workgroup (8 waves) per (sample b, frame f, head h): N patch queries x (1 cls + N patch)
// keys, head dim 64 (timesformer.py:116-131 with the '(b f) n d' grouping :300-301). All keys of the group
// are LDS-resident, so the softmax is exact single-pass (no online rescale):
//   stage   K and V rows -> swizzled row-major LDS images (attn_mfma_common.h): A-fragments of QK^T are
//           ds_read_b128, B-fragments of P.V are ds_read_b64_tr_b16 transpose reads of the SAME kind of image
//           (no transposed copy is ever built); every global load of the staging is issued before the first
//           LDS write (one HBM latency, not seven)
//   S^T = K . Q^T per 16-query tile (v_mfma_f32_16x16x32_bf16; Q fragments straight from HBM, prefetched one
//           tile ahead: each query row is used by exactly one wave). In the C layout every lane owns ONE
//           query column, so max/sum are in-lane plus two xor-shuffles, and the exponentiated tile is already
//           the A operand of O = P.V (k-order permuted identically on the V side): no cross-lane traffic.
//           exp is one v_fma + one v_exp (scale and log2e folded), bf16 packing is v_cvt_pk_bf16_f32.
//   O tile -> per-wave LDS transpose -> 16-B row-contiguous stores.
// The CLS query (token 0) attends to ALL keys; each workgroup runs it as one extra query "tile" through the
// same MFMA path and emits the flash-style partial (max, sum, un-normalised acc[64]) over its own frame's keys
// (the cls key itself is taken by frame 0); a tiny combine kernel merges the F partials: K and V are read from
// HBM once and no separate pass or barrier is spent on the CLS row.
//
// Groups of up to 272 keys (TSF-B/16 and TSF-L/14 at 224) use 8 waves and two workgroups per CU. Larger groups
// (TSF-L/14 at 336: 577 keys) keep the SAME exact single-pass structure: the two images of 592 key rows fill
// 148 KiB of the CU's 160 KiB LDS, one workgroup of 4 waves per CU (one per SIMD, up to 512 registers each: the 37
// score tiles of a query tile stay in registers), no online-softmax rescaling and no second pass over K/V.
//
// PRECISION POLICY (template P, attn_mfma_common.h): PrecBf16 is the benched kernel; PrecSplit is the same kernel text on
// float32 tensors with every operand as hi/lo bf16 images and 3 MFMAs per product (f32-class: the parity configuration
// runs THIS kernel, not the shape-generic one); groups of up to 272 keys (four images fill the LDS beyond that).
//
// Roofline: algorithmic HBM bytes per (b,f,h) = 4 * N * 64 * 2 (q,k,v in, o out); MFMA work is ~1/3 of
// the HBM time at 8 TB/s on TSF-B, so the kernel is built to stream: 2
// workgroups (16 waves) per CU (<= 80 KB LDS each) overlap one group's staging with the other's MFMA phase.
#include "attn_mfma_common.h"

namespace {

using namespace attn_mfma;
constexpr int CLS_REC = 66;       // cls partial record: m, l, acc[64]
constexpr int kBigTiles = 37;     // large-group variant: up to 592 keys (TSF-L/14 at 336: 577)

template <int NKT, int NW, int IMAGES = 1> struct SpaceLds {
  static constexpr int KROWS = NKT * 16;
  static constexpr int ks_off = 0;                                       // bytes
  static constexpr int vs_off = ks_off + KROWS * RS * 2;
  static constexpr int lo_off = 2 * KROWS * RS;                          // ELEMENTS from a hi image to its lo image (PrecSplit)
  static constexpr int ot_off = IMAGES * (vs_off + KROWS * RS * 2);      // NW waves x [16][OS] bf16
  static constexpr int total = ot_off + NW * 16 * OS * 2;
};

// TEXT = true reuses the kernel for the causal text tower (openai_model.py:196-198): one group per (b, h),
// L queries x L keys, no cls row, key j visible to query i iff j <= i, no CLS partial.
// MASKALL: the tile count is an upper bound of ceil(nkeys/16) (every tile is masked), for the large-group variant.
template <typename P, int NKT, bool TEXT, int NW, bool MASKALL>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 1 : ((NKT <= 13 && !P::kSplit) ? 4 : 2))) void space_fwd_kernel(
    const typename P::io_t* __restrict__ qkv, typename P::io_t* __restrict__ out, float* __restrict__ lse,
    float* __restrict__ cls_ws, int F, int N, int H) {
  using io_t = typename P::io_t;
  using Op = typename P::Op;
  using Tr = typename P::Tr;
  constexpr int NT = NW * 64;
  using L = SpaceLds<NKT, NW, P::kImages>;
  constexpr int LO = L::lo_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* Ks = reinterpret_cast<uint16_t*>(smem + L::ks_off);
  uint16_t* Vs = reinterpret_cast<uint16_t*>(smem + L::vs_off);
  uint16_t* Ot = reinterpret_cast<uint16_t*>(smem + L::ot_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % H, f = (blockIdx.x / H) % F, b = blockIdx.x / (H * F);
  const int D = H * 64, T = TEXT ? N : 1 + F * N, nkeys = TEXT ? N : N + 1;
  const size_t tstride = (size_t)3 * D;
  const io_t* base = qkv + (size_t)b * T * tstride + h * 64;          // + token*3D (+D: k, +2D: v)
  const int tok0 = TEXT ? 0 : 1 + f * N;                               // token of query 0 (and of key row 1)
  const int c = lane & 15, g = lane >> 4;
  // query tiles: nqt patch tiles; in the space/time towers one more "tile" holds the CLS query (token 0), whose
  // flash-style partial over this frame's keys comes out of the very same MFMA path (column 0 of that tile)
  const int nqt = (N + 15) / 16, ntiles = TEXT ? nqt : nqt + 1;

  // Q fragments of this wave's first tile go out before the staging loads
  auto q_ptr = [&](int qt) {
    const int qr = qt * 16 + c;
    const int tok = (!TEXT && qt == nqt) ? 0 : tok0 + (qr < N ? qr : N - 1);
    return base + (size_t)tok * tstride + g * 8;
  };
  const int qt_first = wave < ntiles ? wave : 0;
  Op qn0 = P::load_op(q_ptr(qt_first)), qn1 = P::load_op(q_ptr(qt_first) + 32);

  // K and V rows -> LDS images. Key row r is token tok0 + r - 1 (r >= 1) or the cls token (r = 0).
  {
    const io_t* krow0 = base + (size_t)(TEXT ? 0 : tok0 - 1) * tstride + D;
    // at most 8 passes (16 loads per thread) in flight at a time
    constexpr int RPP = NT / 8, GROUP = 8 * RPP;
#pragma unroll 1
    for (int r0 = 0; r0 < L::KROWS; r0 += GROUP) {
      const int pad = L::KROWS - r0 < GROUP ? L::KROWS - r0 : GROUP;
      constexpr int MAXP = (L::KROWS < GROUP ? L::KROWS + RPP - 1 : GROUP) / RPP;
      stage_rows2<P, NT, MAXP>(Ks + r0 * RS, krow0 + (size_t)r0 * tstride, tstride,
                               (TEXT || r0 != 0) ? nullptr : base + D, Vs + r0 * RS, krow0 + D + (size_t)r0 * tstride,
                               tstride, (TEXT || r0 != 0) ? nullptr : base + 2 * D, pad, nkeys - r0, tid, LO);
    }
  }
  __syncthreads();

  constexpr float kScale = 0.125f, kExp2 = 0.125f * 1.4426950408889634f;     // exp(x*scale) = exp2(x*kExp2)
  uint16_t* ot = Ot + wave * 16 * OS;
  const FragOff fo = frag_offsets(lane);
#pragma unroll 1
  for (int qt = wave; qt < ntiles; qt += NW) {
    const bool cls_tile = !TEXT && qt == nqt;
    const int qrow = qt * 16 + c;
    const Op qf0 = qn0, qf1 = qn1;
    if (qt + NW < ntiles) {             // prefetch the next tile's Q fragments under this tile's MFMAs
      qn0 = P::load_op(q_ptr(qt + NW));
      qn1 = P::load_op(q_ptr(qt + NW) + 32);
    }
    // Key tiles are swept in register groups of GS tiles (one group = all tiles for the 8-wave kernels: exact
    // single pass; the large-group variant keeps 12 score tiles in registers at a time and rescales the running
    // (max, sum, O) flash-style between groups -- K and V stay LDS-resident, nothing is re-read). Per group: two
    // sweeps over its tiles (d 0..31, then d 32..63): consecutive MFMAs are independent, the accumulate of a tile
    // is GS issue slots behind its first half, so nothing waits on MFMA latency.
    constexpr int GS = NKT <= 17 ? NKT : 12;
    constexpr int NG = (NKT + GS - 1) / GS;
    float m = -INFINITY, l = 0.f;            // running max (raw score units) and this lane's share of the row sum
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int t0 = gi * GS;                         // first tile of the group (compile-time after unrolling)
      const int gn = NKT - t0 < GS ? NKT - t0 : GS;   // tiles in this group
      f32x4 acc[GS];
#pragma unroll
      for (int k = 0; k < GS; ++k) {
        if (k < gn) acc[k] = mfma(P::tile_op(Ks, LO, t0 + k, fo.a[0]), qf0, f32x4{0.f, 0.f, 0.f, 0.f});
      }
#pragma unroll
      for (int k = 0; k < GS; ++k) {
        if (k < gn) acc[k] = mfma(P::tile_op(Ks, LO, t0 + k, fo.a[1]), qf1, acc[k]);
      }
      // acc[k][r] = raw S[query c][key (t0+k)*16 + g*4 + r]. Space groups: NKT = ceil(nkeys/16) exactly, so only
      // the last tile can hold padded keys (compile-time); text: causal mask on every tile. The CLS query sees
      // the cls key (row 0) only in frame 0, so that the F partials count it once.
      if (gi == 0 && cls_tile && f != 0 && g == 0) acc[0][0] = -INFINITY;
      float mg = -INFINITY;
#pragma unroll
      for (int k = 0; k < GS; ++k) {
        if (k < gn) {
          const int kt = t0 + k;
          if (TEXT || MASKALL || kt == NKT - 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 16 + g * 4 + r;
              const bool vis = key < nkeys && (!TEXT || key <= qrow);
              acc[k][r] = vis ? acc[k][r] : -INFINITY;
            }
          }
          mg = max3_raw(mg, acc[k][0], acc[k][1]);         // (no canonicalising v_max x,x in front of MFMA results)
          mg = max3_raw(mg, acc[k][2], acc[k][3]);
        }
      }
      mg = rows4_max(mg);
      if (NG > 1) {
        // rescale what the earlier groups accumulated; a group (or everything so far) may be fully masked: -inf
        const float mn = fmaxf(m, mg);
        const float al = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m - mn) * kExp2);
        l *= al;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          // o[dt][r] belongs to query g*4+r; al belongs to query c: fetch the factor of the right query
#pragma unroll
          for (int r = 0; r < 4; ++r) o[dt][r] *= __shfl(al, g * 4 + r, 64);
        }
        m = mn;
      } else {
        m = mg;
      }
      const float mk = (m == -INFINITY) ? 0.f : m * kExp2;
      // p = exp2(s * kExp2 - mk) and the row sum on score pairs (v_pk_fma_f32 / v_pk_add_f32)
      const f32x2 sc2 = {kExp2, kExp2}, nm2 = {-mk, -mk};
      f32x2 l2 = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < GS; ++k) {
        if (k < gn) {
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2 e = f32x2{acc[k][r], acc[k][r + 1]} * sc2 + nm2;
            const f32x2 pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            acc[k][r] = pp[0];
            acc[k][r + 1] = pp[1];
            l2 += pp;
          }
        }
      }
      l += l2[0] + l2[1];
#pragma unroll
      for (int j = 0; j < (GS + 1) / 2; ++j) {
        if (2 * j < gn) {
          const bool two = 2 * j + 1 < gn;
          const int j1 = two ? 2 * j + 1 : 2 * j;
          Op pa;
          if (two)
            pa = P::pack(acc[2 * j], acc[j1]);
          else
            pa = P::pack_lo(acc[2 * j]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const Tr lo = P::tile_tr(Vs, LO, t0 + 2 * j, fo.tr[dt]);
            Tr hi = P::zero_tr();
            if (two) hi = P::tile_tr(Vs, LO, t0 + 2 * j + 1, fo.tr[dt]);
            o[dt] = mfma(pa, P::join(lo, hi), o[dt]);
          }
        }
      }
    }
    l = rows4_sum(l);
    // o[dt][r] = O[query g*4+r][d = dt*16 + c]
    if (cls_tile) {
      // record of the CLS query over this frame's keys: (max, sum, un-normalised acc[64]) = column/row 0
      float* rec = cls_ws + (((size_t)b * H + h) * F + f) * CLS_REC;
      if (g == 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) rec[2 + dt * 16 + c] = o[dt][0];
        if (c == 0) { rec[0] = m * kScale; rec[1] = l; }
      }
      continue;
    }
    // normalise, transpose through LDS, store whole rows
    if constexpr (!P::kSplit) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float linv = __builtin_amdgcn_rcpf(__shfl(l, g * 4 + r, 64));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[(g * 4 + r) * OS + dt * 16 + c] = f32_to_bf16(o[dt][r] * linv);
      }
      // same wave wrote and reads: LDS ops of one wave complete in order, no barrier needed
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = (lane >> 3) + 8 * k, ch = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(ot + row * OS + ch * 8);
        const int q = qt * 16 + row;
        if (q < N) *reinterpret_cast<uint4*>(out + ((size_t)b * T + tok0 + q) * D + h * 64 + ch * 8) = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float linv = 1.0f / __shfl(l, g * 4 + r, 64);          // f32 class: a true division
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt][r] *= linv;
      }
      store_tile_rows<P>(ot, o, 1.0f, lane,
                         [&](int row) { return out + ((size_t)b * T + tok0 + qt * 16 + row) * D + h * 64; },
                         [&](int row) { return qt * 16 + row < N; });
    }
    if (g == 0 && qrow < N) lse[((size_t)b * H + h) * T + tok0 + qrow] = m * kScale + __logf(l);
  }
}

// merges the per-chunk partials of the CLS query: out[b,0,h,:] and lse[b,h,0]
template <typename P>
__global__ __launch_bounds__(64) void cls_combine_kernel(const float* __restrict__ cls_ws,
                                                         typename P::io_t* __restrict__ out,
                                                         float* __restrict__ lse, int nparts, int T, int H) {
  const int h = blockIdx.x % H, b = blockIdx.x / H, d = threadIdx.x;
  const float* rec = cls_ws + ((size_t)b * H + h) * nparts * CLS_REC;
  float M = -INFINITY;
  for (int p = 0; p < nparts; ++p) M = fmaxf(M, rec[p * CLS_REC]);
  float Lsum = 0.f, acc = 0.f;
  for (int p = 0; p < nparts; ++p) {
    const float w = __expf(rec[p * CLS_REC] - M);
    Lsum = fmaf(rec[p * CLS_REC + 1], w, Lsum);
    acc = fmaf(rec[p * CLS_REC + 2 + d], w, acc);
  }
  out[(size_t)b * T * H * 64 + h * 64 + d] = P::from_f32(acc / Lsum);
  if (d == 0) lse[((size_t)b * H + h) * T] = M + __logf(Lsum);
}

template <typename P, int NKT, bool TEXT = false, int NW = 8, bool MASKALL = false>
int launch_space_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st) {
  using L = SpaceLds<NKT, NW, P::kImages>;
  using io_t = typename P::io_t;
  static_assert(L::total <= 160 * 1024, "LDS per CU");   // <= 80 KB (NKT <= 13) keeps 2 workgroups per CU
  if (L::total > 64 * 1024)
    if (int rc = lvl_allow_lds<space_fwd_kernel<P, NKT, TEXT, NW, MASKALL>>()) return rc;
  hipLaunchKernelGGL((space_fwd_kernel<P, NKT, TEXT, NW, MASKALL>), dim3((unsigned)(B * F * H)), dim3(NW * 64),
                     L::total, st, (const io_t*)qkv, (io_t*)out, lse, ws, F, N, H);
  LVL_CHECK_LAUNCH("space_fwd_mfma");
  if (TEXT) return LVL_OK;
  hipLaunchKernelGGL(cls_combine_kernel<P>, dim3((unsigned)(B * H)), dim3(64), 0, st, ws, (io_t*)out, lse, F,
                     1 + F * N, H);
  LVL_CHECK_LAUNCH("cls_combine");
  return LVL_OK;
}

template <typename P>
int dispatch_space_fwd_small(int nkeys, const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H,
                             hipStream_t st) {
  switch ((nkeys + 15) / 16) {          // exact tile count: the kernel masks only the last key tile
#define SPACE_FWD_CASE(K) case K: return launch_space_fwd<P, K>(qkv, out, lse, ws, B, F, N, H, st);
    SPACE_FWD_CASE(1) SPACE_FWD_CASE(2) SPACE_FWD_CASE(3) SPACE_FWD_CASE(4) SPACE_FWD_CASE(5) SPACE_FWD_CASE(6)
    SPACE_FWD_CASE(7) SPACE_FWD_CASE(8) SPACE_FWD_CASE(9) SPACE_FWD_CASE(10) SPACE_FWD_CASE(11) SPACE_FWD_CASE(12)
    SPACE_FWD_CASE(13) SPACE_FWD_CASE(14) SPACE_FWD_CASE(15) SPACE_FWD_CASE(16) SPACE_FWD_CASE(17)
#undef SPACE_FWD_CASE
  }
  return 1;       // more than 272 keys
}

template <typename P>
int dispatch_text_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, hipStream_t st) {
  if (L <= 64) return launch_space_fwd<P, 4, true>(qkv, out, lse, nullptr, B, 1, L, H, st);
  if (L <= 128) return launch_space_fwd<P, 8, true>(qkv, out, lse, nullptr, B, 1, L, H, st);
  if (L <= 208) return launch_space_fwd<P, 13, true>(qkv, out, lse, nullptr, B, 1, L, H, st);
  if (L <= 272) return launch_space_fwd<P, 17, true>(qkv, out, lse, nullptr, B, 1, L, H, st);
  return lvl_fail(LVL_ENOSYS, "text_mfma_fwd: context length %d exceeds the LDS-resident kernel", L);
}

}  // namespace

void lvl_launch_cls_combine(const float* ws, void* out, float* lse, int B, int H, int nparts, int T, int dtype,
                            hipStream_t st) {
  if (dtype == LVL_F32)
    hipLaunchKernelGGL(cls_combine_kernel<PrecSplit>, dim3((unsigned)(B * H)), dim3(64), 0, st, ws, (float*)out, lse,
                       nparts, T, H);
  else
    hipLaunchKernelGGL(cls_combine_kernel<PrecBf16>, dim3((unsigned)(B * H)), dim3(64), 0, st, ws, (uint16_t*)out, lse,
                       nparts, T, H);
}

// bf16: groups of up to 592 keys; float32 (f32-class, PrecSplit: four LDS images): up to 272 keys
bool lvl_space_mfma_supported(int F, int N, int dtype) {
  return N + 1 <= (dtype == LVL_F32 ? 17 : kBigTiles) * 16 && N >= 1 && F <= 64;
}

int lvl_space_mfma_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, int dtype,
                       hipStream_t st) {
  const int nkeys = N + 1;
  if (dtype == LVL_F32) {
    const int rc = dispatch_space_fwd_small<PrecSplit>(nkeys, qkv, out, lse, ws, B, F, N, H, st);
    return rc == 1 ? lvl_fail(LVL_ENOSYS, "space_mfma_fwd (f32 class): %d keys per group exceed the LDS", nkeys) : rc;
  }
  const int rc = dispatch_space_fwd_small<PrecBf16>(nkeys, qkv, out, lse, ws, B, F, N, H, st);
  if (rc != 1) return rc;
  // large groups: 4 waves, up to 592 keys resident; the exact-tile-count instantiation for 577..592 keys
  // (TSF-L/14 at 336), every tile masked otherwise
  if ((nkeys + 15) / 16 == kBigTiles)
    return launch_space_fwd<PrecBf16, kBigTiles, false, 4, false>(qkv, out, lse, ws, B, F, N, H, st);
  if (nkeys <= kBigTiles * 16)
    return launch_space_fwd<PrecBf16, kBigTiles, false, 4, true>(qkv, out, lse, ws, B, F, N, H, st);
  return lvl_fail(LVL_ENOSYS, "space_mfma_fwd: %d keys per group exceeds the LDS-resident kernel", nkeys);
}

bool lvl_text_mfma_supported(int L) { return L >= 1 && L <= 272; }

int lvl_text_mfma_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int dtype, hipStream_t st) {
  if (dtype == LVL_F32) return dispatch_text_fwd<PrecSplit>(qkv, out, lse, B, L, H, st);
  return dispatch_text_fwd<PrecBf16>(qkv, out, lse, B, L, H, st);
}
