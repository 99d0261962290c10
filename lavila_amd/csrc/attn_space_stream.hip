// Space-mode divided attention for LARGE groups (TSF-L/14 at 336: 576 patches + cls = 577 keys per frame), forward and
// backward, as KEY-TILED STREAMING kernels on the matrix cores, gfx950.
//
// The resident kernels (attn_space_mfma.hip / attn_space_bwd.hip) keep every key of a (sample, frame, head) group in
// LDS; at 577 keys the two images fill 148 KiB, so one 4-wave workgroup owns a compute unit and the kernels run at 0.12
// of HBM / 0.11 of the MFMA peak -- latency bound (profiles/r03_fp8_qk_and_config4_attention.txt). Here the group is cut
// along the rows a wave OWNS (queries in the forward and dQ kernels, keys in the dK/dV kernel): a workgroup of 4 waves
// owns 128 of them (two 16-row MFMA tiles per wave, fragments straight from HBM, each LDS fragment read feeds two MFMAs)
// and STREAMS the other side through a double-buffered pair of 64-row LDS images (32 KiB): the next chunk's global loads
// are in flight while the current chunk is multiplied, one barrier per chunk, 3-4 workgroups (12-16 waves) per compute
// unit. The other side is re-read once per 128 own rows -- 5 times per group at 577 keys -- but from the L2 (148 KB per
// group), not from HBM.
//   forward  online softmax (running max / sum, rescale per chunk), O^T accumulated transposed (channels x queries) so
//            that the rescale factor and the final 1/l are lane-local and a lane stores 4 consecutive channels directly;
//            the cls query rides as one more query tile and leaves the usual per-frame partial (cls_combine_kernel).
//   dq       P recomputed from the saved lse; dS^T packed straight into the B operand of dQ^T += K^T dS^T; delta = dO.O
//            per query written for the dK/dV kernel; the cls query's partial dQ goes to the f32 atomic slab.
//   dkv      waves own 32 keys, Q / dO (+ lse, delta) stream; the cls query is query row N of the group; the cls key's
//            dK / dV go to the atomic slab (it collects gradient from every frame).
// Same precision policies as the resident kernels (attn_mfma_common.h): PrecBf16, and PrecSplit for float32 tensors
// (f32-class: hi/lo images, 3 MFMAs per product), which also takes the 577-key float32 groups off the generic kernels.
#include "attn_mfma_common.h"

namespace {

using namespace attn_mfma;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kExp2 = 0.125f * kLog2e;          // exp(s * scale) = exp2(s * kExp2)
constexpr int CLS_REC = 66;                       // cls partial record: m, l, acc[64]
constexpr int KC = 64;                            // streamed rows per chunk (4 MFMA tiles)

template <int IMAGES> struct StreamLds {
  static constexpr int img_elems = KC * RS;                      // one 64-row image
  static constexpr int buf_elems = 2 * img_elems;                // a buffer = image A | image B
  static constexpr int lo_off = 2 * buf_elems;                   // ELEMENTS from a hi image to its lo image (PrecSplit)
  static constexpr int img_bytes = IMAGES * 2 * buf_elems * 2;   // two buffers, hi (+ lo)
  static constexpr int vec_off = img_bytes;                      // dkv kernel: lse | delta of the two buffers' rows
  static constexpr int total_fwd = img_bytes;
  static constexpr int total_dkv = img_bytes + 2 * 2 * KC * 4;
  static constexpr int total_dma = 3 * buf_elems * 2;            // LDS-DMA ring of three stages (bf16)
  static constexpr int total_dma_dkv = total_dma + 3 * 4 * KC * 4;   // + lse / delta rows of the three stages
};

// One thread's share of a 64-row chunk: rows r_in and r_in + 32 of both images, 8 channels at c8.
template <typename P>
struct ChunkRegs { typename P::Raw a[2], b[2]; };

// ---- LDS-DMA ring (bf16 instantiations) -------------------------------------------------------------------------
// With register staging the next chunk's loads are issued when the current chunk starts and must have landed when it
// ends: ONE chunk of run-ahead, eight staging registers per thread and a ds_write pass per chunk. The
// DMA form keeps a ring of NST = 3 stages filled by global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass):
// chunk c+2 is requested before chunk c is multiplied. A stage is 16 fills of 1 KiB (8 rows x 128 B; fills 0-7 image A,
// 8-15 image B), four per wave; lane l of a fill lands at row 8*blk + (l >> 3), PHYSICAL slot l & 7, and therefore
// fetches the logical slot (l & 7) ^ (row & 7) of its row (the images' XOR swizzle, applied on the source side).
// Rows past the end of the group re-read the last valid row (never used unmasked). The fills go through inline asm:
// hipcc would otherwise drain them (vmcnt(0)) at every barrier; the waits are counted by hand (4 fills per wave and
// chunk, in-order completion). Measured at 16 x 577 keys, batch 8 (profiles/r04_config4_attention_dma_ring.txt): forward
// 0.417 -> 0.404 ms, forward + backward 1.355 -> 1.335 ms against register staging (variant bit 2) -- the chunks were NOT
// waiting for memory (the hypothesis this was built on); what bounds them is the softmax's VALU work.
constexpr int NST = 3;

#pragma clang diagnostic ignored "-Winline-asm"
template <typename RowPtr>
__device__ __forceinline__ void dma_issue_chunk(uint16_t* stage, int ch, int nrows, int wave_u, int lane, RowPtr row_ptr) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = wave_u * 4 + i;
    const int image = f >> 3, blk = f & 7;
    const int row = blk * 8 + (lane >> 3);
    int idx = ch * KC + row;
    idx = idx < nrows ? idx : nrows - 1;
    const uint16_t* src = row_ptr(idx, image) + (((lane & 7) ^ (row & 7)) << 3);
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(stage + image * (KC * RS)) + (uint32_t)blk * 1024u);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory", "m0");
  }
}
// The common case of a fill: chunk rows that are consecutive tokens and all inside the group. The per-lane part of the
// address (row within the 8-row fill, swizzled slot) is the same for every fill of every chunk -- one 32-bit VGPR offset
// computed once -- and everything else rides in the scalar base: no vector arithmetic per fill.
//   voff  = (lane >> 3) * row_bytes + (((lane & 7) ^ ((lane >> 3) & 7)) << 4)
//   base0 = first byte of row (wave's first 8-row block) of the chunk in the wave's image; + i * 8 rows per fill
__device__ __forceinline__ void dma_issue_linear(uint16_t* stage, int wave_u, const char* base0, uint32_t voff,
                                                 uint32_t rows8_bytes) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = wave_u * 4 + i;
    const int image = f >> 3, blk = f & 7;
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(stage + image * (KC * RS)) + (uint32_t)blk * 1024u);
    const char* b = base0 + (size_t)i * rows8_bytes;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(b)
                 : "memory", "m0");
  }
}
// forces the compiler to have the operand's registers loaded here (an empty asm that "modifies" them)
__device__ __forceinline__ void pin_op(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void pin_op(Op2& v) {
  pin_op(v.h);
  pin_op(v.l);
}
__device__ __forceinline__ void pin_f32(float& v) { asm volatile("" : "+v"(v)); }

// wait for this wave's fills of the chunk about to be multiplied (the next chunk's four may stay in flight), then meet
// the other waves: everybody's fills of the chunk have landed and everybody is done with the previous chunk
template <int FILLS = 4>
__device__ __forceinline__ void dma_wait_chunk(bool more_in_flight) {
  if (more_in_flight)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FILLS) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
// OCC = workgroups per compute unit the register budget is cut for (bf16: 3 -> <= 168 VGPRs, 4 -> <= 128 with a few
// spilled dwords; which one runs is a measured choice, lvl_debug_stream_variant)
template <typename P, int OCC, bool DMA>
__global__ __launch_bounds__(256, (P::kSplit ? 1 : OCC)) void space_stream_fwd_kernel(
    const typename P::io_t* __restrict__ qkv, typename P::io_t* __restrict__ out, float* __restrict__ lse,
    float* __restrict__ cls_ws, int F, int N, int H, int NB, int NG) {
  using io_t = typename P::io_t;
  using Op = typename P::Op;
  using Tr = typename P::Tr;
  using L = StreamLds<P::kImages>;
  constexpr int LO = L::lo_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* img = reinterpret_cast<uint16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware decode: workgroup i runs on XCD i % 8 (observed, for speed only); the NB workgroups of a group take the
  // same XCD so that the rows they all stream are fetched into ONE L2 instead of up to NB of them (measured at 577
  // keys with group-major block order: the streamed side crossed the fabric ~5x)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int blk = slot % NB, grp = (slot / NB) * 8 + xcd;
  if (grp >= NG) return;
  const int h = grp % H, f = (grp / H) % F, b = grp / (H * F);
  const int D = H * 64, T = 1 + F * N, nkeys = N + 1, tok0 = 1 + f * N;
  const size_t ts = (size_t)3 * D;
  const io_t* base = qkv + (size_t)b * T * ts + h * 64;
  const int c = lane & 15, g = lane >> 4;
  const int nqt = (N + 15) / 16, ntiles = nqt + 1;          // patch query tiles + the cls query tile
  const FragOff fo = frag_offsets(lane);

  // this wave's two query tiles; a tile beyond the group aliases tile 0 (computed, never stored)
  int qt[2];
  bool live[2], cls_t[2];
  Op qf[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qt[t] = blk * 8 + wave * 2 + t;
    live[t] = qt[t] < ntiles;
    if (!live[t]) qt[t] = 0;
    cls_t[t] = qt[t] == nqt;
    const int qr = qt[t] * 16 + c;
    const int tok = cls_t[t] ? 0 : tok0 + (qr < N ? qr : N - 1);
    const io_t* qp = base + (size_t)tok * ts + g * 8;
    qf[t][0] = P::load_op(qp);
    qf[t][1] = P::load_op(qp + 32);
  }

  // chunk staging: key row kidx = chunk * KC + r is the cls token (kidx = 0) or token tok0 + kidx - 1
  const int c8 = tid & 7, r_in = tid >> 3;
  ChunkRegs<P> cr;
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int kidx = ch * KC + p * 32 + r_in;
      cr.a[p] = P::zero_raw();
      cr.b[p] = P::zero_raw();
      if (kidx < nkeys) {
        const io_t* kp = base + (size_t)(kidx == 0 ? 0 : tok0 + kidx - 1) * ts + D + c8 * 8;
        cr.a[p] = P::load_raw(kp);
        cr.b[p] = P::load_raw(kp + D);
      }
    }
  };
  auto store_chunk = [&](int buf) {
    uint16_t* ka = img + buf * L::buf_elems;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      P::stage(ka, LO, img_off(p * 32 + r_in, c8), cr.a[p]);
      P::stage(ka + L::img_elems, LO, img_off(p * 32 + r_in, c8), cr.b[p]);
    }
  };

  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f32x4 o[2][4];                                  // O^T: [channel dt*16 + g*4 + r][query c]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = (nkeys + KC - 1) / KC;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto key_row = [&](int kidx, int image) {          // K (image 0) / V (image 1) row of key kidx, as bf16 elements
    return reinterpret_cast<const uint16_t*>(base) + (size_t)(kidx == 0 ? 0 : tok0 + kidx - 1) * ts + D * (1 + image);
  };
  // ring depth: three stages (two chunks of run-ahead) at 3 workgroups per CU; the 4-workgroup cut has LDS for two
  // stages (4 x 32 KiB): the next chunk is requested behind the barrier that frees its stage and has one whole chunk of
  // arithmetic to land
  constexpr int NS = OCC >= 4 ? 2 : NST;
  int stage = 0;
  if constexpr (DMA) {
    dma_issue_chunk(img, 0, nkeys, wave_u, lane, key_row);
    if (NS == 3 && nchunks > 1) dma_issue_chunk(img + L::buf_elems, 1, nkeys, wave_u, lane, key_row);
    // Pin the query fragments NOW. hipcc defers the wait for their loads to the first use INSIDE the loop, where its
    // s_waitcnt vmcnt(0) (it does not count the asm fills) would drain the run-ahead fills in every iteration.
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) pin_op(qf[t][hh]);
  } else {
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
  }
  // linear fills (chunks >= 2 that lie inside the group): the lane's constant offset and the wave's scalar row base
  const uint32_t row_bytes = (uint32_t)(ts * sizeof(io_t));
  const uint32_t dma_voff = (uint32_t)(lane >> 3) * row_bytes + (uint32_t)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
  const char* dma_row0 = reinterpret_cast<const char*>(base + (size_t)(tok0 - 1) * ts + D * (1 + (wave_u >> 1))) +
                         (size_t)((wave_u & 1) * 32) * row_bytes;          // key row 0 would sit here (rows >= 1 do)
#pragma unroll 1
  for (int ch = 0; ch < nchunks; ++ch) {
    if constexpr (DMA) {
      dma_wait_chunk(NS == 3 && ch + 1 < nchunks);
      constexpr int AHEAD = NS - 1;
      if (ch + AHEAD < nchunks) {      // into the stage chunk ch-1 used: every wave has passed the barrier behind it
        const int s2 = stage == 0 ? NS - 1 : stage - 1;
        if ((ch + AHEAD + 1) * KC <= nkeys)    // every row of that chunk is a key of the group: no clamp, no cls row
          dma_issue_linear(img + s2 * L::buf_elems, wave_u, dma_row0 + (size_t)((ch + AHEAD) * KC) * row_bytes, dma_voff,
                           8u * row_bytes);
        else
          dma_issue_chunk(img + s2 * L::buf_elems, ch + AHEAD, nkeys, wave_u, lane, key_row);
      }
    } else {
      if (ch + 1 < nchunks) load_chunk(ch + 1);
      stage = ch & 1;
    }
    const uint16_t* Ks = img + stage * L::buf_elems;
    const uint16_t* Vs = Ks + L::img_elems;
    const int k0 = ch * KC;
    const int nt = (nkeys - k0 + 15) / 16 < 4 ? (nkeys - k0 + 15) / 16 : 4;      // key tiles of this chunk (uniform)
    f32x4 s[2][4];
    if (k0 + KC <= nkeys) {
      // a full chunk (all but the last one or two): eight fragment reads up front, sixteen MFMAs, no masks
      Op ka[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ka[k][0] = P::tile_op(Ks, LO, k, fo.a[0]);
        ka[k][1] = P::tile_op(Ks, LO, k, fo.a[1]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) s[t][k] = P::mfma_qk(ka[k][0], qf[t][0], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t) s[t][k] = P::mfma_qk(ka[k][1], qf[t][1], s[t][k]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < nt) {
          const Op ka0 = P::tile_op(Ks, LO, k, fo.a[0]), ka1 = P::tile_op(Ks, LO, k, fo.a[1]);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            s[t][k] = P::mfma_qk(ka0, qf[t][0], f32x4{0.f, 0.f, 0.f, 0.f});
            s[t][k] = P::mfma_qk(ka1, qf[t][1], s[t][k]);
          }
        } else {
          s[0][k] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          s[1][k] = s[0][k];
        }
      }
      // the chunk holds padded keys (zero rows): mask them
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s[t][k][r] = (k0 + k * 16 + g * 4 + r < nkeys) ? s[t][k][r] : -INFINITY;
    }
    // s[t][k][r] = raw S[query c of tile t][key k0 + k*16 + g*4 + r]
    Op pa[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // the cls query sees the cls key (key 0) only in frame 0, so that the F partials count it once
      if (ch == 0 && cls_t[t] && f != 0 && g == 0) s[t][0][0] = -INFINITY;
      float mg = max3_raw(s[t][0][0], s[t][0][1], s[t][0][2]);         // 16 scores: eight three-way maxima
#pragma unroll
      for (int i = 3; i < 15; i += 2) mg = max3_raw(mg, s[t][i >> 2][i & 3], s[t][(i + 1) >> 2][(i + 1) & 3]);
      mg = fmaxf(mg, s[t][3][3]);
      mg = rows4_max(mg);
      const float mn = fmaxf(m[t], mg);
      const float al = (m[t] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m[t] - mn) * kExp2);
      const float mk = (mn == -INFINITY) ? 0.f : mn * kExp2;
      m[t] = mn;
      // p = exp2(s * kExp2 - mk), two scores per packed multiply-add; the row sum as packed adds
      const f32x2 sc2 = {kExp2, kExp2}, nm2 = {-mk, -mk};
      f32x2 ls2 = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 e = f32x2{s[t][k][r], s[t][k][r + 1]} * sc2 + nm2;
          const f32x2 pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
          s[t][k][r] = pp[0];
          s[t][k][r + 1] = pp[1];
          ls2 += pp;
        }
      l[t] = l[t] * al + (ls2[0] + ls2[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][dt][r] *= al;           // transposed accumulator: the factor is lane-local
      pa[t][0] = P::pack(s[t][0], s[t][1]);
      pa[t][1] = P::pack(s[t][2], s[t][3]);
    }
    // O^T += V^T P^T over the chunk's two 32-key halves: every transpose read of V feeds both query tiles
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (2 * j < nt) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const Tr lo = P::tile_tr(Vs, LO, 2 * j, fo.tr[dt]);
          const Tr hi = P::tile_tr(Vs, LO, 2 * j + 1, fo.tr[dt]);
          const Op vb = P::join(lo, hi);
          o[0][dt] = mfma(vb, pa[0][j], o[0][dt]);
          o[1][dt] = mfma(vb, pa[1][j], o[1][dt]);
        }
      }
    }
    if constexpr (DMA) {
      stage = stage == NS - 1 ? 0 : stage + 1;
    } else {
      if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);
      __syncthreads();
    }
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float lt = rows4_sum(l[t]);
    if (!live[t]) continue;
    if (cls_t[t]) {
      // record of the cls query over this frame's keys: (max, sum, un-normalised acc[64]) = query column 0
      float* rec = cls_ws + (((size_t)b * H + h) * F + f) * CLS_REC;
      if (c == 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) rec[2 + dt * 16 + g * 4 + r] = o[t][dt][r];
        if (g == 0) { rec[0] = m[t] * 0.125f; rec[1] = lt; }
      }
      continue;
    }
    const int qrow = qt[t] * 16 + c;
    if (qrow < N) {
      const float linv = P::kSplit ? 1.0f / lt : __builtin_amdgcn_rcpf(lt);
      io_t* orow = out + ((size_t)b * T + tok0 + qrow) * D + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        P::store4(orow + dt * 16 + g * 4, o[t][dt][0] * linv, o[t][dt][1] * linv, o[t][dt][2] * linv,
                  o[t][dt][3] * linv);
      if (g == 0) lse[((size_t)b * H + h) * T + tok0 + qrow] = m[t] * 0.125f + __logf(lt);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// dQ (and delta)
// ------------------------------------------------------------------------------------------------------------
template <typename P, bool DMA>
__global__ __launch_bounds__(256, (P::kSplit ? 1 : 3)) void space_stream_dq_kernel(
    const typename P::io_t* __restrict__ qkv, const typename P::io_t* __restrict__ out,
    const typename P::io_t* __restrict__ dout, const float* __restrict__ lse, typename P::io_t* __restrict__ dqkv,
    float* __restrict__ delta, float* __restrict__ atom_ws, int F, int N, int H, int NB, int NG) {
  using io_t = typename P::io_t;
  using Op = typename P::Op;
  using Tr = typename P::Tr;
  using L = StreamLds<P::kImages>;
  constexpr int LO = L::lo_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* img = reinterpret_cast<uint16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware decode: workgroup i runs on XCD i % 8 (observed, for speed only); the NB workgroups of a group take the
  // same XCD so that the rows they all stream are fetched into ONE L2 instead of up to NB of them (measured at 577
  // keys with group-major block order: the streamed side crossed the fabric ~5x)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int blk = slot % NB, grp = (slot / NB) * 8 + xcd;
  if (grp >= NG) return;
  const int h = grp % H, f = (grp / H) % F, b = grp / (H * F);
  const int D = H * 64, T = 1 + F * N, nkeys = N + 1, tok0 = 1 + f * N;
  const size_t ts = (size_t)3 * D;
  const io_t* base = qkv + (size_t)b * T * ts + h * 64;
  const io_t* obase = out + (size_t)b * T * D + h * 64;
  const io_t* dobase = dout + (size_t)b * T * D + h * 64;
  const float* lrow = lse + ((size_t)b * H + h) * T;
  float* drow = delta + ((size_t)b * H + h) * T;
  float* cls_slab = atom_ws + (((size_t)b * H + h) * F + f) * 192;        // d cls q | d cls k | d cls v
  const int c = lane & 15, g = lane >> 4;
  const int nqt = (N + 15) / 16, ntiles = nqt + 1;
  const FragOff fo = frag_offsets(lane);

  int qt[2];
  bool live[2], cls_t[2];
  Op qf[2][2], gf[2][2];
  float Lk[2], dl[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qt[t] = blk * 8 + wave * 2 + t;
    live[t] = qt[t] < ntiles;
    if (!live[t]) qt[t] = 0;
    cls_t[t] = qt[t] == nqt;
    const int qr = qt[t] * 16 + c;
    const int tok = cls_t[t] ? 0 : tok0 + (qr < N ? qr : N - 1);
    const io_t* qp = base + (size_t)tok * ts + g * 8;
    qf[t][0] = P::load_op(qp);
    qf[t][1] = P::load_op(qp + 32);
    gf[t][0] = P::load_op(dobase + (size_t)tok * D + g * 8);
    gf[t][1] = P::load_op(dobase + (size_t)tok * D + g * 8 + 32);
    const Op y0 = P::load_op(obase + (size_t)tok * D + g * 8), y1 = P::load_op(obase + (size_t)tok * D + g * 8 + 32);
    float a[8], bb[8], acc = 0.f;
    P::to_f32(gf[t][0], a);
    P::to_f32(y0, bb);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(a[i], bb[i], acc);
    P::to_f32(gf[t][1], a);
    P::to_f32(y1, bb);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(a[i], bb[i], acc);
    acc = rows4_sum(acc);
    dl[t] = acc;
    Lk[t] = lrow[tok] * kLog2e;
    // delta of the patch queries and -- once per group: it is the same in every frame -- of the cls query (token 0)
    if (live[t] && g == 0 && ((!cls_t[t] && qr < N) || (cls_t[t] && c == 0))) drow[tok] = acc;
  }

  const int c8 = tid & 7, r_in = tid >> 3;
  ChunkRegs<P> cr;
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int kidx = ch * KC + p * 32 + r_in;
      cr.a[p] = P::zero_raw();
      cr.b[p] = P::zero_raw();
      if (kidx < nkeys) {
        const io_t* kp = base + (size_t)(kidx == 0 ? 0 : tok0 + kidx - 1) * ts + D + c8 * 8;
        cr.a[p] = P::load_raw(kp);
        cr.b[p] = P::load_raw(kp + D);
      }
    }
  };
  auto store_chunk = [&](int buf) {
    uint16_t* ka = img + buf * L::buf_elems;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      P::stage(ka, LO, img_off(p * 32 + r_in, c8), cr.a[p]);
      P::stage(ka + L::img_elems, LO, img_off(p * 32 + r_in, c8), cr.b[p]);
    }
  };

  f32x4 o[2][4];                                  // dQ^T: [channel dt*16 + g*4 + r][query c]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = (nkeys + KC - 1) / KC;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto key_row = [&](int kidx, int image) {
    return reinterpret_cast<const uint16_t*>(base) + (size_t)(kidx == 0 ? 0 : tok0 + kidx - 1) * ts + D * (1 + image);
  };
  int stage = 0;
  if constexpr (DMA) {
    dma_issue_chunk(img, 0, nkeys, wave_u, lane, key_row);
    if (nchunks > 1) dma_issue_chunk(img + L::buf_elems, 1, nkeys, wave_u, lane, key_row);
#pragma unroll
    for (int t = 0; t < 2; ++t) {                  // see the forward kernel: no deferred compiler waits inside the loop
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) { pin_op(qf[t][hh]); pin_op(gf[t][hh]); }
      pin_f32(Lk[t]);
      pin_f32(dl[t]);
    }
  } else {
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
  }
  // linear fills (see the forward kernel): chunks whose rows are all keys of the group need no per-fill vector arithmetic
  const uint32_t row_bytes = (uint32_t)(ts * sizeof(io_t));
  const uint32_t dma_voff = (uint32_t)(lane >> 3) * row_bytes + (uint32_t)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
  const char* dma_row0 = reinterpret_cast<const char*>(base + (size_t)(tok0 - 1) * ts + D * (1 + (wave_u >> 1))) +
                         (size_t)((wave_u & 1) * 32) * row_bytes;
#pragma unroll 1
  for (int ch = 0; ch < nchunks; ++ch) {
    if constexpr (DMA) {
      dma_wait_chunk(ch + 1 < nchunks);
      if (ch + 2 < nchunks) {
        const int s2 = stage == 0 ? 2 : stage - 1;
        if ((ch + 3) * KC <= nkeys)
          dma_issue_linear(img + s2 * L::buf_elems, wave_u, dma_row0 + (size_t)((ch + 2) * KC) * row_bytes, dma_voff,
                           8u * row_bytes);
        else
          dma_issue_chunk(img + s2 * L::buf_elems, ch + 2, nkeys, wave_u, lane, key_row);
      }
    } else {
      if (ch + 1 < nchunks) load_chunk(ch + 1);
      stage = ch & 1;
    }
    const uint16_t* Ks = img + stage * L::buf_elems;
    const uint16_t* Vs = Ks + L::img_elems;
    const int k0 = ch * KC;
    const int nt = (nkeys - k0 + 15) / 16 < 4 ? (nkeys - k0 + 15) / 16 : 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {                 // the chunk's two 32-key halves
      if (2 * j >= nt) continue;
      Op kf[2][2], vf[2][2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          kf[kt][hh] = P::tile_op(Ks, LO, 2 * j + kt, fo.a[hh]);
          vf[kt][hh] = P::tile_op(Vs, LO, 2 * j + kt, fo.a[hh]);
        }
      Op pa[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        f32x4 p0 = {-dl[t], -dl[t], -dl[t], -dl[t]}, p1 = p0;     // dP - delta: delta rides in the accumulator
        s0 = P::mfma_qk(kf[0][0], qf[t][0], s0);
        p0 = mfma(vf[0][0], gf[t][0], p0);
        s1 = P::mfma_qk(kf[1][0], qf[t][0], s1);
        p1 = mfma(vf[1][0], gf[t][0], p1);
        s0 = P::mfma_qk(kf[0][1], qf[t][1], s0);
        p0 = mfma(vf[0][1], gf[t][1], p0);
        s1 = P::mfma_qk(kf[1][1], qf[t][1], s1);
        p1 = mfma(vf[1][1], gf[t][1], p1);
        float d0[4], d1[4];
        const f32x2 k2 = {kExp2, kExp2}, nl2 = {-Lk[t], -Lk[t]};
#pragma unroll
        for (int r = 0; r < 4; r += 2) {            // score pairs: packed multiply-add / multiply
          const f32x2 a0 = f32x2{s0[r], s0[r + 1]} * k2 + nl2, a1 = f32x2{s1[r], s1[r + 1]} * k2 + nl2;
          f32x2 e0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
          f32x2 e1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int key0 = k0 + (2 * j) * 16 + g * 4 + r + u, key1 = key0 + 16;
            e0[u] = key0 < nkeys ? e0[u] : 0.f;     // padded key rows are zero, but exp2(-lse) may overflow: mask
            e1[u] = key1 < nkeys ? e1[u] : 0.f;
            if (cls_t[t] && f != 0 && key0 == 0) e0[u] = 0.f;      // (cls query, cls key) outside frame 0
          }
          const f32x2 x0 = e0 * f32x2{p0[r], p0[r + 1]}, x1 = e1 * f32x2{p1[r], p1[r + 1]};
          d0[r] = x0[0]; d0[r + 1] = x0[1];
          d1[r] = x1[0]; d1[r + 1] = x1[1];
        }
        pa[t] = P::pack(d0, d1);
      }
      // dQ^T += K^T . dS^T for these 32 keys: A fragments are transpose reads of the K image
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const Tr lo = P::tile_tr(Ks, LO, 2 * j, fo.tr[dt]);
        const Tr hi = P::tile_tr(Ks, LO, 2 * j + 1, fo.tr[dt]);
        const Op kb = P::join(lo, hi);
        o[0][dt] = mfma(kb, pa[0], o[0][dt]);
        o[1][dt] = mfma(kb, pa[1], o[1][dt]);
      }
    }
    if constexpr (DMA) {
      stage = stage == NST - 1 ? 0 : stage + 1;
    } else {
      if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);
      __syncthreads();
    }
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!live[t]) continue;
    const int qrow = qt[t] * 16 + c;
    if (cls_t[t]) {
      if (c == 0) {                               // this frame's share of d(cls q): slot f of the partial slab (one writer)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          *reinterpret_cast<float4*>(cls_slab + dt * 16 + g * 4) =
              make_float4(o[t][dt][0] * 0.125f, o[t][dt][1] * 0.125f, o[t][dt][2] * 0.125f, o[t][dt][3] * 0.125f);
      }
    } else if (qrow < N) {
      io_t* row = dqkv + (size_t)b * T * ts + (size_t)(tok0 + qrow) * ts + h * 64;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        P::store4(row + dt * 16 + g * 4, o[t][dt][0] * 0.125f, o[t][dt][1] * 0.125f, o[t][dt][2] * 0.125f,
                  o[t][dt][3] * 0.125f);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// dK / dV
// ------------------------------------------------------------------------------------------------------------
template <typename P, bool DMA>
__global__ __launch_bounds__(256, (P::kSplit ? 1 : 2)) void space_stream_dkv_kernel(
    const typename P::io_t* __restrict__ qkv, const typename P::io_t* __restrict__ dout,
    const float* __restrict__ lse, const float* __restrict__ delta, typename P::io_t* __restrict__ dqkv,
    float* __restrict__ atom_ws, int F, int N, int H, int NB, int NG) {
  using io_t = typename P::io_t;
  using Op = typename P::Op;
  using Tr = typename P::Tr;
  using L = StreamLds<P::kImages>;
  constexpr int LO = L::lo_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* img = reinterpret_cast<uint16_t*>(smem);
  // register staging: [2 buffers][lse(64, log2 units) | delta(64)] behind the images; DMA ring: [NST stages][4][64] raw
  // lse | delta | (the same again: waves 2 and 3 issue the duplicates so that every wave has 5 fills per chunk)
  float* vec = reinterpret_cast<float*>(smem + (DMA ? L::total_dma : L::vec_off));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware decode: workgroup i runs on XCD i % 8 (observed, for speed only); the NB workgroups of a group take the
  // same XCD so that the rows they all stream are fetched into ONE L2 instead of up to NB of them (measured at 577
  // keys with group-major block order: the streamed side crossed the fabric ~5x)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int blk = slot % NB, grp = (slot / NB) * 8 + xcd;
  if (grp >= NG) return;
  const int h = grp % H, f = (grp / H) % F, b = grp / (H * F);
  const int D = H * 64, T = 1 + F * N, nkeys = N + 1, tok0 = 1 + f * N;
  const int nq = N + 1;                             // queries of the group: N patch rows, then the cls query (row N)
  const size_t ts = (size_t)3 * D;
  const io_t* base = qkv + (size_t)b * T * ts + h * 64;
  const io_t* dobase = dout + (size_t)b * T * D + h * 64;
  const float* lrow = lse + ((size_t)b * H + h) * T;
  const float* drow = delta + ((size_t)b * H + h) * T;
  float* cls_slab = atom_ws + (((size_t)b * H + h) * F + f) * 192;
  const int c = lane & 15, g = lane >> 4;
  const int nkt = (nkeys + 15) / 16;
  const FragOff fo = frag_offsets(lane);

  // this wave's two key tiles (32 keys): fragments straight from HBM
  int kt[2];
  bool live[2];
  Op kk[2][2], vv[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    kt[t] = blk * 8 + wave * 2 + t;
    live[t] = kt[t] < nkt;
    const int krow = kt[t] * 16 + c;
    kk[t][0] = P::zero_op(); kk[t][1] = kk[t][0]; vv[t][0] = kk[t][0]; vv[t][1] = kk[t][0];
    if (live[t] && krow < nkeys) {
      const io_t* kptr = base + (size_t)(krow == 0 ? 0 : tok0 + krow - 1) * ts + D + g * 8;
      kk[t][0] = P::load_op(kptr);
      kk[t][1] = P::load_op(kptr + 32);
      vv[t][0] = P::load_op(kptr + D);
      vv[t][1] = P::load_op(kptr + D + 32);
    }
  }

  // chunk staging: query row qidx = chunk * KC + r is patch token tok0 + qidx (qidx < N) or the cls token (qidx = N)
  const int c8 = tid & 7, r_in = tid >> 3;
  ChunkRegs<P> cr;
  float lse_r = INFINITY, del_r = 0.f;              // threads 0..63: lse (log2 units) / delta of row tid of the chunk
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int qidx = ch * KC + p * 32 + r_in;
      cr.a[p] = P::zero_raw();
      cr.b[p] = P::zero_raw();
      if (qidx < nq) {
        const int tk = qidx < N ? tok0 + qidx : 0;
        cr.a[p] = P::load_raw(base + (size_t)tk * ts + c8 * 8);
        cr.b[p] = P::load_raw(dobase + (size_t)tk * D + c8 * 8);
      }
    }
    if (tid < KC) {
      const int qidx = ch * KC + tid;
      lse_r = INFINITY;                             // padded queries: exp2(-inf) = 0
      del_r = 0.f;
      if (qidx < nq) {
        const int tk = qidx < N ? tok0 + qidx : 0;
        lse_r = lrow[tk] * kLog2e;
        del_r = drow[tk];
      }
    }
  };
  auto store_chunk = [&](int buf) {
    uint16_t* qa = img + buf * L::buf_elems;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      P::stage(qa, LO, img_off(p * 32 + r_in, c8), cr.a[p]);
      P::stage(qa + L::img_elems, LO, img_off(p * 32 + r_in, c8), cr.b[p]);
    }
    if (tid < KC) {
      vec[buf * 2 * KC + tid] = lse_r;
      vec[buf * 2 * KC + KC + tid] = del_r;
    }
  };

  f32x4 adk[2][4], adv[2][4];                       // dK^T, dV^T: [channel dt*16 + g*4 + r][key c]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { adk[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; adv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nchunks = (nq + KC - 1) / KC;
  const int cls_chunk = N / KC, cls_half = (N % KC) >> 5, cls_sub = N & 31;     // where the cls query sits
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto query_row = [&](int qidx, int image) {       // Q (image 0) / dO (image 1) row of query qidx (row N = the cls query)
    const int tk = qidx < N ? tok0 + qidx : 0;
    return image == 0 ? reinterpret_cast<const uint16_t*>(base) + (size_t)tk * ts
                      : reinterpret_cast<const uint16_t*>(dobase) + (size_t)tk * D;
  };
  // fifth fill of a chunk: the 64 lse (waves 0, 2) or delta (waves 1, 3) values of its rows, one float per lane
  auto issue_vec = [&](int st, int ch) {
    int qidx = ch * KC + lane;
    qidx = qidx < nq ? qidx : nq - 1;
    const int tk = qidx < N ? tok0 + qidx : 0;
    const float* src = ((wave_u & 1) ? drow : lrow) + tk;
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(vec + (st * 4 + wave_u) * KC));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v), "v"(src) : "memory", "m0");
  };
  int stage = 0;
  if constexpr (DMA) {
    dma_issue_chunk(img, 0, nq, wave_u, lane, query_row);
    issue_vec(0, 0);
    if (nchunks > 1) {
      dma_issue_chunk(img + L::buf_elems, 1, nq, wave_u, lane, query_row);
      issue_vec(1, 1);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)                    // no deferred compiler waits inside the loop (see the forward kernel)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) { pin_op(kk[t][hh]); pin_op(vv[t][hh]); }
  } else {
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
  }
  // linear fills: chunks of patch queries only (the cls query, row N, sits in the last chunk). Waves 0, 1 stream Q rows
  // (stride 3D), waves 2, 3 dO rows (stride D); the lse / delta fill is 64 consecutive floats
  const uint32_t row_bytes = (uint32_t)(((wave_u >> 1) ? (size_t)D : ts) * sizeof(io_t));
  const uint32_t dma_voff = (uint32_t)(lane >> 3) * row_bytes + (uint32_t)(((lane & 7) ^ ((lane >> 3) & 7)) << 4);
  const char* dma_row0 = ((wave_u >> 1) ? reinterpret_cast<const char*>(dobase + (size_t)tok0 * D)
                                        : reinterpret_cast<const char*>(base + (size_t)tok0 * ts)) +
                         (size_t)((wave_u & 1) * 32) * row_bytes;
  const char* vec_row0 = reinterpret_cast<const char*>(((wave_u & 1) ? drow : lrow) + tok0);
#pragma unroll 1
  for (int ch = 0; ch < nchunks; ++ch) {
    if constexpr (DMA) {
      dma_wait_chunk<5>(ch + 1 < nchunks);
      if (ch + 2 < nchunks) {
        const int s2 = stage == 0 ? 2 : stage - 1;
        if ((ch + 3) * KC <= N) {
          dma_issue_linear(img + s2 * L::buf_elems, wave_u, dma_row0 + (size_t)((ch + 2) * KC) * row_bytes, dma_voff,
                           8u * row_bytes);
          const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(vec + (s2 * 4 + wave_u) * KC));
          const char* vb = vec_row0 + (size_t)((ch + 2) * KC) * 4;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(m0v), "v"((uint32_t)lane * 4u),
                       "s"(vb) : "memory", "m0");
        } else {
          dma_issue_chunk(img + s2 * L::buf_elems, ch + 2, nq, wave_u, lane, query_row);
          issue_vec(s2, ch + 2);
        }
      }
    } else {
      if (ch + 1 < nchunks) load_chunk(ch + 1);
      stage = ch & 1;
    }
    const uint16_t* Qs = img + stage * L::buf_elems;
    const uint16_t* Gs = Qs + L::img_elems;
    const float* lse_s = DMA ? vec + stage * 4 * KC : vec + stage * 2 * KC;
    const float* del_s = lse_s + KC;
    const int q0 = ch * KC;
    const int nqt_c = (nq - q0 + 15) / 16 < 4 ? (nq - q0 + 15) / 16 : 4;       // query tiles of this chunk (uniform)
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {               // the chunk's two 32-query halves
      if (2 * qh >= nqt_c) continue;
      const uint16_t* Qp = Qs + qh * 32 * RS;
      const uint16_t* Gp = Gs + qh * 32 * RS;
      Op qa[2][2], ga[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          qa[q][hh] = P::tile_op(Qp, LO, q, fo.a[hh]);
          ga[q][hh] = P::tile_op(Gp, LO, q, fo.a[hh]);
        }
      const float4 ls0 = *reinterpret_cast<const float4*>(lse_s + qh * 32 + g * 4);
      const float4 ls1 = *reinterpret_cast<const float4*>(lse_s + qh * 32 + 16 + g * 4);
      const float4 de0 = *reinterpret_cast<const float4*>(del_s + qh * 32 + g * 4);
      const float4 de1 = *reinterpret_cast<const float4*>(del_s + qh * 32 + 16 + g * 4);
      float lsa[8] = {ls0.x, ls0.y, ls0.z, ls0.w, ls1.x, ls1.y, ls1.z, ls1.w};
      const float dea[8] = {de0.x, de0.y, de0.z, de0.w, de1.x, de1.y, de1.z, de1.w};
      if constexpr (DMA) {
        // the ring holds the raw lse of (clamped) rows: log2 units here; padded queries (last chunk) -> exp2(-inf) = 0
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int qidx = q0 + qh * 32 + (r >> 2) * 16 + g * 4 + (r & 3);
          lsa[r] = qidx < nq ? lsa[r] * kLog2e : INFINITY;
        }
      }
      // (cls query, cls key) outside frame 0 is not attended: one element of the group's first key tile
      const bool kill_here = f != 0 && ch == cls_chunk && qh == cls_half;
      Op pa[2], da[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        f32x4 p0 = {-dea[0], -dea[1], -dea[2], -dea[3]}, p1 = {-dea[4], -dea[5], -dea[6], -dea[7]};    // dP - delta
        s0 = P::mfma_qk(qa[0][0], kk[t][0], s0);
        s1 = P::mfma_qk(qa[1][0], kk[t][0], s1);
        p0 = mfma(ga[0][0], vv[t][0], p0);
        p1 = mfma(ga[1][0], vv[t][0], p1);
        s0 = P::mfma_qk(qa[0][1], kk[t][1], s0);
        s1 = P::mfma_qk(qa[1][1], kk[t][1], s1);
        p0 = mfma(ga[0][1], vv[t][1], p0);
        p1 = mfma(ga[1][1], vv[t][1], p1);
        // s0[r] = S[query q0 + qh*32 + g*4 + r][key kt[t]*16 + c], s1: queries + 16
        const bool kill_pair = kill_here && kt[t] == 0 && c == 0;
        float e0[4], e1[4], d0[4], d1[4];
        const f32x2 k2 = {kExp2, kExp2};
#pragma unroll
        for (int r = 0; r < 4; r += 2) {            // score pairs: packed multiply-add / multiply
          const f32x2 a0 = f32x2{s0[r], s0[r + 1]} * k2 - f32x2{lsa[r], lsa[r + 1]};
          const f32x2 a1 = f32x2{s1[r], s1[r + 1]} * k2 - f32x2{lsa[4 + r], lsa[5 + r]};
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            e0[r + u] = __builtin_amdgcn_exp2f(a0[u]);
            e1[r + u] = __builtin_amdgcn_exp2f(a1[u]);
            e0[r + u] = (kill_pair && g * 4 + r + u == cls_sub) ? 0.f : e0[r + u];
            e1[r + u] = (kill_pair && 16 + g * 4 + r + u == cls_sub) ? 0.f : e1[r + u];
          }
          const f32x2 x0 = f32x2{e0[r], e0[r + 1]} * f32x2{p0[r], p0[r + 1]};
          const f32x2 x1 = f32x2{e1[r], e1[r + 1]} * f32x2{p1[r], p1[r + 1]};
          d0[r] = x0[0]; d0[r + 1] = x0[1];
          d1[r] = x1[0]; d1[r + 1] = x1[1];
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
    if constexpr (DMA) {
      stage = stage == NST - 1 ? 0 : stage + 1;
    } else {
      if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);
      __syncthreads();
    }
  }

  io_t* dkb = dqkv + (size_t)b * T * ts + D + h * 64;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!live[t]) continue;
    const int krow = kt[t] * 16 + c;
    if (krow >= 1 && krow < nkeys) {
      io_t* row = dkb + (size_t)(tok0 + krow - 1) * ts;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        P::store4(row + dt * 16 + g * 4, adk[t][dt][0] * 0.125f, adk[t][dt][1] * 0.125f, adk[t][dt][2] * 0.125f,
                  adk[t][dt][3] * 0.125f);
        P::store4(row + D + dt * 16 + g * 4, adv[t][dt][0], adv[t][dt][1], adv[t][dt][2], adv[t][dt][3]);
      }
    } else if (krow == 0) {          // the cls KEY collects gradient from every frame: this frame's slot (one writer)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<float4*>(cls_slab + 64 + dt * 16 + g * 4) =
            make_float4(adk[t][dt][0] * 0.125f, adk[t][dt][1] * 0.125f, adk[t][dt][2] * 0.125f, adk[t][dt][3] * 0.125f);
        *reinterpret_cast<float4*>(cls_slab + 128 + dt * 16 + g * 4) =
            make_float4(adv[t][dt][0], adv[t][dt][1], adv[t][dt][2], adv[t][dt][3]);
      }
    }
  }
}

std::atomic<int> g_stream_mode{0};      // 0 auto (groups the resident kernels do not take two-per-CU), 1 always, -1 never
// bf16 kernels. bit 0: the forward's 3-workgroup cut with a three-stage LDS-DMA ring (default: 4 workgroups per CU, two
// stages); bit 2: register staging instead of the LDS-DMA rings (forward: bit 0 then picks its 4-workgroup cut)
std::atomic<int> g_stream_variant{0};

template <typename P>
int launch_stream_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st) {
  using io_t = typename P::io_t;
  using L = StreamLds<P::kImages>;
  const int NB = ((N + 15) / 16 + 1 + 7) / 8;
  const int NG = B * F * H;                       // groups; the grid is padded to whole rounds of 8 XCDs
  const dim3 grid((unsigned)((NG + 7) / 8 * 8 * NB));
  if constexpr (std::is_same<P, PrecFp8QK>::value) {
    // one cut: LDS-DMA ring, 3 workgroups per CU (the fragment conversions do not fit the 128-register cut)
    hipLaunchKernelGGL((space_stream_fwd_kernel<P, 3, true>), grid, dim3(256), L::total_dma, st, (const io_t*)qkv,
                       (io_t*)out, lse, ws, F, N, H, NB, NG);
  } else if constexpr (P::kSplit) {
    if (L::total_fwd > 64 * 1024)
      if (int rc = lvl_allow_lds<space_stream_fwd_kernel<P, 3, false>>()) return rc;
    hipLaunchKernelGGL((space_stream_fwd_kernel<P, 3, false>), grid, dim3(256), L::total_fwd, st, (const io_t*)qkv,
                       (io_t*)out, lse, ws, F, N, H, NB, NG);
  } else {
    const int v = g_stream_variant.load();
    if (!(v & 4)) {
      // default: LDS-DMA, the 4-workgroup cut (two-stage ring, 128 registers: 0.359 ms against 0.367 at 16 x 577 keys,
      // batch 8); bit 0: the 3-workgroup cut with the three-stage ring
      if (v & 1)
        hipLaunchKernelGGL((space_stream_fwd_kernel<P, 3, true>), grid, dim3(256), L::total_dma, st, (const io_t*)qkv,
                           (io_t*)out, lse, ws, F, N, H, NB, NG);
      else
        hipLaunchKernelGGL((space_stream_fwd_kernel<P, 4, true>), grid, dim3(256), L::total_fwd, st, (const io_t*)qkv,
                           (io_t*)out, lse, ws, F, N, H, NB, NG);
    } else if (v & 1) {
      hipLaunchKernelGGL((space_stream_fwd_kernel<P, 4, false>), grid, dim3(256), L::total_fwd, st, (const io_t*)qkv,
                         (io_t*)out, lse, ws, F, N, H, NB, NG);
    } else {
      hipLaunchKernelGGL((space_stream_fwd_kernel<P, 3, false>), grid, dim3(256), L::total_fwd, st, (const io_t*)qkv,
                         (io_t*)out, lse, ws, F, N, H, NB, NG);
    }
  }
  LVL_CHECK_LAUNCH("space_stream_fwd");
  return LVL_OK;
}

template <typename P>
int launch_stream_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                      float* atom_ws, int B, int F, int N, int H, hipStream_t st) {
  using io_t = typename P::io_t;
  using L = StreamLds<P::kImages>;
  const int NBq = ((N + 15) / 16 + 1 + 7) / 8, NBk = ((N + 1 + 15) / 16 + 7) / 8;
  if constexpr (P::kSplit) {
    if (L::total_dkv > 64 * 1024) {
      if (int rc = lvl_allow_lds<space_stream_dq_kernel<P, false>>()) return rc;
      if (int rc = lvl_allow_lds<space_stream_dkv_kernel<P, false>>()) return rc;
    }
  }
  const int NG = B * F * H;
  const dim3 gq((unsigned)((NG + 7) / 8 * 8 * NBq)), gk((unsigned)((NG + 7) / 8 * 8 * NBk));
  if constexpr (!P::kSplit) {
    // default: LDS-DMA rings (the fp8 QK^T policy has no other cut)
    if (std::is_same<P, PrecFp8QK>::value || !(g_stream_variant.load() & 4)) {
      hipLaunchKernelGGL((space_stream_dq_kernel<P, true>), gq, dim3(256), L::total_dma, st, (const io_t*)qkv,
                         (const io_t*)out, (const io_t*)dout, lse, (io_t*)dqkv, delta, atom_ws, F, N, H, NBq, NG);
      LVL_CHECK_LAUNCH("space_stream_dq");
      hipLaunchKernelGGL((space_stream_dkv_kernel<P, true>), gk, dim3(256), L::total_dma_dkv, st, (const io_t*)qkv,
                         (const io_t*)dout, lse, delta, (io_t*)dqkv, atom_ws, F, N, H, NBk, NG);
      LVL_CHECK_LAUNCH("space_stream_dkv");
      return LVL_OK;
    }
  }
  if constexpr (!std::is_same<P, PrecFp8QK>::value) {
    hipLaunchKernelGGL((space_stream_dq_kernel<P, false>), gq, dim3(256), L::total_fwd, st, (const io_t*)qkv,
                       (const io_t*)out, (const io_t*)dout, lse, (io_t*)dqkv, delta, atom_ws, F, N, H, NBq, NG);
    LVL_CHECK_LAUNCH("space_stream_dq");
    hipLaunchKernelGGL((space_stream_dkv_kernel<P, false>), gk, dim3(256), L::total_dkv, st, (const io_t*)qkv,
                       (const io_t*)dout, lse, delta, (io_t*)dqkv, atom_ws, F, N, H, NBk, NG);
    LVL_CHECK_LAUNCH("space_stream_dkv");
  }
  return LVL_OK;
}

}  // namespace

void lvl_launch_cls_combine(const float* ws, void* out, float* lse, int B, int H, int nparts, int T, int dtype,
                            hipStream_t st);
void lvl_launch_cls_grad_finalize(const float* atom_ws, void* dqkv, int B, int T, int H, int nslots, int dtype,
                                  hipStream_t st);

// Test / measurement hook: 1 = the streaming kernels for EVERY space group (parity tests at small shapes, A/B against the
// resident kernels), -1 = never, 0 = the shipped choice (lvl_space_stream_wanted).
extern "C" int lvl_debug_space_stream(int mode) {
  g_stream_mode.store(mode < 0 ? -1 : (mode > 0 ? 1 : 0));
  return LVL_OK;
}

extern "C" int lvl_debug_stream_variant(int v) {
  g_stream_variant.store(v);
  return LVL_OK;
}

// the shipped choice: groups of more than 288 keys (bf16: the resident kernels would run one 4-wave workgroup per CU;
// float32: their four images do not fit the LDS at all)
// fp8 QK^T (PrecFp8QK): -1 = not decided yet (LAVILA_FP8_QK in the environment, default off), 0 / 1 = off / on
std::atomic<int> g_fp8_qk{-1};
static bool fp8_qk() {
  int v = g_fp8_qk.load();
  if (v < 0) {
    const char* e = getenv("LAVILA_FP8_QK");
    v = (e != nullptr && e[0] == '1') ? 1 : 0;
    g_fp8_qk.store(v);
  }
  return v == 1;
}
extern "C" int lvl_set_fp8_qk(int on) {
  g_fp8_qk.store(on ? 1 : 0);
  return LVL_OK;
}

bool lvl_space_stream_wanted(int F, int N, int dtype) {
  const int mode = g_stream_mode.load();
  if (mode != 0) return mode > 0;
  return N + 1 > 288 || (dtype == LVL_F32 && N + 1 > 272);
}

int lvl_space_stream_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, int dtype,
                         hipStream_t st) {
  const int rc = dtype == LVL_F32 ? launch_stream_fwd<PrecSplit>(qkv, out, lse, ws, B, F, N, H, st)
                 : fp8_qk()       ? launch_stream_fwd<PrecFp8QK>(qkv, out, lse, ws, B, F, N, H, st)
                                  : launch_stream_fwd<PrecBf16>(qkv, out, lse, ws, B, F, N, H, st);
  if (rc != LVL_OK) return rc;
  lvl_launch_cls_combine(ws, out, lse, B, H, F, 1 + F * N, dtype, st);
  LVL_CHECK_LAUNCH("cls_combine");
  return LVL_OK;
}

// ws layout: delta [B*H*T] f32, then atomics [B*H*192] f32 (d cls q | d cls k | d cls v)
int lvl_space_stream_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* ws,
                         int B, int F, int N, int H, int dtype, hipStream_t st) {
  const int T = 1 + F * N;
  float* delta = ws;
  float* atom_ws = ws + (size_t)B * H * T;            // partial records [B*H][F][192]: every slot has its writer, no zeroing
  const int rc = dtype == LVL_F32
                     ? launch_stream_bwd<PrecSplit>(qkv, out, dout, lse, dqkv, delta, atom_ws, B, F, N, H, st)
                 : fp8_qk() ? launch_stream_bwd<PrecFp8QK>(qkv, out, dout, lse, dqkv, delta, atom_ws, B, F, N, H, st)
                            : launch_stream_bwd<PrecBf16>(qkv, out, dout, lse, dqkv, delta, atom_ws, B, F, N, H, st);
  if (rc != LVL_OK) return rc;
  lvl_launch_cls_grad_finalize(atom_ws, dqkv, B, T, H, F, dtype, st);
  LVL_CHECK_LAUNCH("cls_grad_finalize");
  return LVL_OK;
}
