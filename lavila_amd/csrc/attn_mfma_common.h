// Shared pieces of the MFMA attention kernels (forward + backward), gfx950.
//
// LDS images are ROW-MAJOR with 128-byte rows (64 bf16 = one head row) and a 16-byte-slot XOR swizzle
//     phys_slot = slot ^ (row & 7)
// One image serves both operand kinds of a 16x16x32 MFMA:
//   * contraction over d (QK^T, dO V^T): fragment = 8 consecutive channels of one row -> one ds_read_b128
//     (conflict-free: the 16 rows of a lane group land on 16 different slots);
//   * contraction over rows (P V, dS K, dS^T Q, P^T dO): fragment = 4 consecutive ROWS at one channel -> one
//     ds_read_b64_tr_b16 (the gfx950 LDS transpose read: within a 16-lane group, lane m points at row
//     r0 + m/4, channels d0 + 4*(m%4) .. +3, and lane c receives rows r0..r0+3 at channel d0 + c; verified on
//     hardware by tools/probes/tr_read_probe.hip). No transposed copies are ever staged.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace attn_mfma {

constexpr int RS = 64;   // image row stride in elements (128 B)
constexpr int OS = 72;   // per-wave output transposition tile stride (elements)

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) { return f32x2_to_bf16x2(lo, hi); }
__device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
}

// element offset of 16-byte slot `slot` (8 channels) of row `row`
__device__ __forceinline__ int img_off(int row, int slot) { return row * RS + ((slot ^ (row & 7)) << 3); }

// 8 consecutive channels [slot*8, slot*8+8) of one row
__device__ __forceinline__ uint4 img_frag(const uint16_t* img, int row, int slot) {
  return *reinterpret_cast<const uint4*>(img + img_off(row, slot));
}

// rows r0..r0+3 at channel d0 + (lane & 15); r0 must be the same for the 16 lanes of a group (it may differ
// between groups), d0 a multiple of 16
__device__ __forceinline__ uint2 img_frag_tr(const uint16_t* img, int r0, int d0, int lane) {
  const int m = lane & 15, row = r0 + (m >> 2), col = d0 + ((m & 3) << 2);
  const uint16_t* p = img + row * RS + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

// Per-lane fragment offsets inside one 16-row tile (tiles start at multiples of 16 rows, so the swizzle term is
// tile-invariant): computed once, every access is then `image + tile*16*RS + offset` (an immediate for unrolled
// tile loops).
struct FragOff {
  int a[2];     // ds_read_b128 fragment of row (lane & 15): channels ks*32 + (lane>>4)*8 .. +7, ks = 0, 1
  int tr[4];    // transpose-read fragment: rows (lane>>4)*4 .. +3 at channel dt*16 + (lane & 15), dt = 0..3
};
__device__ __forceinline__ FragOff frag_offsets(int lane) {
  FragOff f;
  const int c = lane & 15, g = lane >> 4, m = c;
  f.a[0] = c * RS + ((g ^ (c & 7)) << 3);
  f.a[1] = c * RS + (((4 + g) ^ (c & 7)) << 3);
  const int rsub = g * 4 + (m >> 2);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    f.tr[dt] = rsub * RS + ((((2 * dt + ((m & 3) >> 1)) ^ (rsub & 7)) << 3) | ((m & 1) << 2));
  return f;
}
__device__ __forceinline__ uint4 tile_frag(const uint16_t* img, int tile, int off) {
  return *reinterpret_cast<const uint4*>(img + tile * 16 * RS + off);
}
__device__ __forceinline__ uint2 tile_frag_tr(const uint16_t* img, int tile, int off) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(img + tile * 16 * RS + off));
  return __builtin_bit_cast(uint2, v);
}

// Cooperative staging of two row sets A and B (`nrows` rows of 64 bf16 each) into swizzled row-major images;
// rows in [nrows, rows_pad) are zero-filled. NT threads, 8 lanes per row, MAXP >= ceil(rows_pad / (NT/8))
// passes. Row r of set X lives at pX + r * strideX (elements), except row 0 when p0X != nullptr (the cls token
// in front of a frame's patch rows). One pointer per thread + a constant stride per pass keeps the address
// arithmetic out of the way, and ALL global loads are issued before the first LDS write so that the passes
// overlap in flight instead of paying one HBM latency each.
template <int NT, int MAXP>
__device__ __forceinline__ void stage_rows2(uint16_t* imgA, const uint16_t* pA, size_t strideA, const uint16_t* p0A,
                                            uint16_t* imgB, const uint16_t* pB, size_t strideB, const uint16_t* p0B,
                                            int rows_pad, int nrows, int tid) {
  constexpr int RPP = NT / 8;
  const int c8 = tid & 7, r_in = tid >> 3;
  const uint16_t* ra = pA + (size_t)r_in * strideA + c8 * 8;
  const uint16_t* rb = pB + (size_t)r_in * strideB + c8 * 8;
  uint4 va[MAXP], vb[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int r = p * RPP + r_in;
    const bool first = p == 0 && r_in == 0;
    const uint16_t* sa = (first && p0A != nullptr) ? p0A + c8 * 8 : ra + (size_t)p * RPP * strideA;
    const uint16_t* sb = (first && p0B != nullptr) ? p0B + c8 * 8 : rb + (size_t)p * RPP * strideB;
    va[p] = make_uint4(0, 0, 0, 0);
    vb[p] = make_uint4(0, 0, 0, 0);
    if (r < nrows) {
      va[p] = *reinterpret_cast<const uint4*>(sa);
      vb[p] = *reinterpret_cast<const uint4*>(sb);
    }
  }
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int r = p * RPP + r_in;
    if (r < rows_pad) {
      *reinterpret_cast<uint4*>(imgA + img_off(r, c8)) = va[p];
      *reinterpret_cast<uint4*>(imgB + img_off(r, c8)) = vb[p];
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
