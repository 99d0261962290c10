// Shared pieces of the MFMA attention kernels (forward + backward), gfx950.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace attn_mfma {

constexpr int KS = 80;   // row-major LDS row stride (elements, 160 B): conflict-free ds_read_b128 fragments
constexpr int OS = 72;   // per-wave output transposition tile stride (elements)

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return f32x2_to_bf16x2(lo, hi); }
__device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
}

// LDS write half of the staging: row r (8 lanes x 16 B, this lane holds chunk c8 = v) into the row-major image
// rm[r*KS + d] and/or the transposed image tr[d*LD + r]. The transposed image is written as packed row pairs
// (lane pairs exchange halves with one xor-8 shuffle) with a per-lane rotation of the write order so that the
// 8 chunk-lanes of a row hit different banks at every step (2-way conflicts at most).
__device__ __forceinline__ void stage_write(uint16_t* rm, uint16_t* tr, int LD, int rows_pad, int r, int c8, uint4 v) {
  const int par = r & 1, rot = c8 & 3;
  if (rm != nullptr && r < rows_pad) *reinterpret_cast<uint4*>(rm + r * KS + c8 * 8) = v;
  if (tr != nullptr) {
    const uint32_t s0 = par ? v.x : v.z, s1 = par ? v.y : v.w;
    const uint32_t p0 = __shfl_xor(s0, 8, 64), p1 = __shfl_xor(s1, 8, 64);
    const uint32_t o0 = par ? v.z : v.x, o1 = par ? v.w : v.y;
    const uint32_t lo0 = par ? p0 : o0, lo1 = par ? p1 : o1;          // even row's two dwords
    const uint32_t hi0 = par ? o0 : p0, hi1 = par ? o1 : p1;          // odd row's two dwords
    const uint32_t pk0 = (lo0 & 0xffffu) | (hi0 << 16), pk1 = (lo0 >> 16) | (hi0 & 0xffff0000u);
    const uint32_t pk2 = (lo1 & 0xffffu) | (hi1 << 16), pk3 = (lo1 >> 16) | (hi1 & 0xffff0000u);
    const uint32_t t0 = (rot & 1) ? pk1 : pk0, t1 = (rot & 1) ? pk2 : pk1, t2 = (rot & 1) ? pk3 : pk2,
                   t3 = (rot & 1) ? pk0 : pk3;
    const uint32_t w0 = (rot & 2) ? t2 : t0, w1 = (rot & 2) ? t3 : t1, w2 = (rot & 2) ? t0 : t2,
                   w3 = (rot & 2) ? t1 : t3;
    if (r < rows_pad) {
      uint16_t* col = tr + (size_t)(c8 * 8 + 4 * par) * LD + (r & ~1);
      *reinterpret_cast<uint32_t*>(col + ((0 + rot) & 3) * LD) = w0;
      *reinterpret_cast<uint32_t*>(col + ((1 + rot) & 3) * LD) = w1;
      *reinterpret_cast<uint32_t*>(col + ((2 + rot) & 3) * LD) = w2;
      *reinterpret_cast<uint32_t*>(col + ((3 + rot) & 3) * LD) = w3;
    }
  }
}

// Cooperative staging of two row sets A and B (`nrows` rows of 64 bf16 each; row r at srcA(r) / srcB(r)); rows in
// [nrows, rows_pad) are zero-filled. NT threads, 8 lanes per row, MAXP >= ceil(rows_pad / (NT/8)) passes.
// ALL global loads are issued before the first LDS write so that the passes overlap in flight instead of
// paying one HBM latency each.
template <int NT, int MAXP, typename SrcA, typename SrcB>
__device__ __forceinline__ void stage_rows2(uint16_t* rmA, uint16_t* trA, SrcA srcA, uint16_t* rmB, uint16_t* trB,
                                            SrcB srcB, int LD, int rows_pad, int nrows, int tid) {
  constexpr int RPP = NT / 8;
  const int c8 = tid & 7, r_in = tid >> 3;
  uint4 va[MAXP], vb[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int r = p * RPP + r_in;
    va[p] = make_uint4(0, 0, 0, 0);
    vb[p] = make_uint4(0, 0, 0, 0);
    if (r < nrows) {
      va[p] = *reinterpret_cast<const uint4*>(srcA(r) + c8 * 8);
      vb[p] = *reinterpret_cast<const uint4*>(srcB(r) + c8 * 8);
    }
  }
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (p * RPP < rows_pad) {           // wave-uniform
      const int r = p * RPP + r_in;
      stage_write(rmA, trA, LD, rows_pad, r, c8, va[p]);
      stage_write(rmB, trB, LD, rows_pad, r, c8, vb[p]);
    }
  }
}

// writes a 16x64 f32 tile held in the MFMA C layout (o[dt][r] = X[row g*4+r][col dt*16+c]) as bf16 rows:
// row i of the tile goes to dst(i) (64 contiguous bf16) if valid(i). Per-wave LDS scratch `ot` ([16][OS]).
template <typename DstFn, typename ValidFn>
__device__ __forceinline__ void store_tile_rows(uint16_t* ot, const f32x4 (&o)[4], float mul, int lane, DstFn dst,
                                                ValidFn valid) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[(g * 4 + r) * OS + dt * 16 + c] = f32_to_bf16(o[dt][r] * mul);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = (lane >> 3) + 8 * k, ch = lane & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(ot + row * OS + ch * 8);
    if (valid(row)) *reinterpret_cast<uint4*>(dst(row) + ch * 8) = v;
  }
}

}  // namespace attn_mfma
