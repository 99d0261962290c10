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

// Reductions over the four 16-lane rows of a wave (the C layout of a 16x16 MFMA tile spreads one query / key column over
// lanes c, c+16, c+32, c+48): two gfx950 row swaps (v_permlane16_swap, v_permlane32_swap: plain VALU, no LDS crossbar
// round trip like the ds_bpermute a __shfl_xor compiles to) and every lane holds the result.
// max of three. fmaxf puts a canonicalising v_max x,x in front of every value the compiler cannot prove quiet -- every
// MFMA result -- unless the translation unit is compiled with -fno-honor-nans (lavila_amd/build.py does that for the
// attention kernels: their scores are never NaN -- finite operands, -inf only through the masks, and every inf - inf is
// guarded by a select). NOT inline asm: the hazard recogniser does not look into asm, and a VALU read of an MFMA result
// needs wait states it would then not insert (measured: NaNs in the split-operand kernels).
__device__ __forceinline__ float max3_raw(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rows4_max(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rows4_sum(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
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

// ---------------------------------------------------------------------------------------------------------------
// Precision policies. Every MFMA attention kernel is a template over one of them; the kernel text -- group decode,
// staging order, tile loops, masks, softmax, cls handling, stores -- is shared, only the operand plumbing differs.
//   PrecBf16   the benched path: bf16 tensors, one bf16 LDS image per operand, one MFMA per product.
//   PrecSplit  the f32-class path (the parity configuration: float32 tensors, north_star "within 1e-3 fp32"): every
//              operand x is carried as TWO bf16 images h = bf16(x), l = bf16(x - h) (x = h + l to 2^-18 |x|), the lo
//              image `lo_off` elements behind the hi image in LDS, and every product is three MFMAs into the same f32
//              accumulator: l.h' + h.l' + h.h' (~2^-17 relative per product, the partial products exact). Probabilities
//              and score gradients, computed in f32 registers, are split the same way before they become operands.
//              Inputs and outputs are float32.
// ---------------------------------------------------------------------------------------------------------------
struct Op2 { uint4 h, l; };      // split operand: 8 contraction elements, hi and lo parts
struct Tr2 { uint2 h, l; };      // split half operand from a transpose read (4 elements)
struct Raw2 { float4 a, b; };    // 8 consecutive float32 channels as loaded from HBM

__device__ __forceinline__ f32x4 mfma(const Op2& a, const Op2& b, f32x4 c) {
  c = mfma(a.l, b.h, c);           // small terms first
  c = mfma(a.h, b.l, c);
  return mfma(a.h, b.h, c);
}

__device__ __forceinline__ void split8(const float (&x)[8], uint4& h, uint4& l) {
  float r[8];
  uint32_t hw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hw[i] = f32x2_to_bf16x2(x[2 * i], x[2 * i + 1]);
    r[2 * i] = x[2 * i] - __uint_as_float(hw[i] << 16);
    r[2 * i + 1] = x[2 * i + 1] - __uint_as_float(hw[i] & 0xffff0000u);
  }
  h = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  l = make_uint4(f32x2_to_bf16x2(r[0], r[1]), f32x2_to_bf16x2(r[2], r[3]), f32x2_to_bf16x2(r[4], r[5]),
                 f32x2_to_bf16x2(r[6], r[7]));
}

// bf16 x 8 fragment -> OCP e4m3 x 8 (gfx950's fp8), saturating at +-448: the operand of v_mfma_f32_16x16x32_fp8_fp8 holds
// the same 8 contraction elements per lane as the bf16 instruction's, in 8 bytes
__device__ __forceinline__ uint2 bf16x8_to_fp8x8(const uint4& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  float f[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __builtin_amdgcn_fmed3f(__uint_as_float(w[i] << 16), -448.f, 448.f);
    f[2 * i + 1] = __builtin_amdgcn_fmed3f(__uint_as_float(w[i] & 0xffff0000u), -448.f, 448.f);
  }
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return make_uint2((uint32_t)lo, (uint32_t)hi);
}
__device__ __forceinline__ f32x4 mfma_fp8(const uint2& a, const uint2& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}

struct PrecBf16 {
  static constexpr bool kSplit = false;
  // the QK^T products of the streaming kernels go through this hook (PrecFp8QK overrides it)
  static __device__ __forceinline__ f32x4 mfma_qk(const uint4& a, const uint4& b, f32x4 c) { return mfma(a, b, c); }
  static constexpr int kImages = 1;       // LDS images per staged operand
  using io_t = uint16_t;
  using Op = uint4;
  using Tr = uint2;
  using Raw = uint4;
  static __device__ __forceinline__ Raw load_raw(const io_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ Raw zero_raw() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ Op op_of(const Raw& r) { return r; }
  static __device__ __forceinline__ Op load_op(const io_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ Op zero_op() { return make_uint4(0, 0, 0, 0); }
  static __device__ __forceinline__ void stage(uint16_t* img, int, int off, const Raw& r) {
    *reinterpret_cast<uint4*>(img + off) = r;
  }
  static __device__ __forceinline__ Op tile_op(const uint16_t* img, int, int tile, int off) {
    return tile_frag(img, tile, off);
  }
  static __device__ __forceinline__ Tr tile_tr(const uint16_t* img, int, int tile, int off) {
    return tile_frag_tr(img, tile, off);
  }
  static __device__ __forceinline__ Tr zero_tr() { return make_uint2(0, 0); }
  static __device__ __forceinline__ Op join(const Tr& lo, const Tr& hi) { return make_uint4(lo.x, lo.y, hi.x, hi.y); }
  // 8 f32 values (a: contraction elements 0..3, b: 4..7) -> operand; A = float[4] or f32x4
  template <typename A>
  static __device__ __forceinline__ Op pack(const A& a, const A& b) {
    return make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3]));
  }
  template <typename A>
  static __device__ __forceinline__ Op pack_lo(const A& a) {          // elements 4..7 zero
    return make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), 0u, 0u);
  }
  static __device__ __forceinline__ void to_f32(const Op& o, float (&v)[8]) {
    Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&o), v);
  }
  static __device__ __forceinline__ float to_f32_1(io_t v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ io_t from_f32(float v) { return f32_to_bf16(v); }
  static __device__ __forceinline__ void store4(io_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
  }
};

// BASELINE configs[3] names an "fp8 MFMA QK^T path" for TSF-L/14 at 336: PrecBf16 with the score products on the fp8
// matrix instruction. q and k fragments are rounded to e4m3 in registers (tensors, LDS images and every other product
// stay bf16; P V, dP, dQ, dK, dV are bf16 MFMAs as before); the backward kernels recompute P through the same hook, so
// the saved lse and the recomputed scores are consistent. An accuracy / speed trade measured in DESIGN.md section 4:
// opt-in (LAVILA_FP8_QK=1 / lvl_set_fp8_qk), streaming kernels only.
struct PrecFp8QK : PrecBf16 {
  static __device__ __forceinline__ f32x4 mfma_qk(const uint4& a, const uint4& b, f32x4 c) {
    return mfma_fp8(bf16x8_to_fp8x8(a), bf16x8_to_fp8x8(b), c);
  }
};

struct PrecSplit {
  static constexpr bool kSplit = true;
  static __device__ __forceinline__ f32x4 mfma_qk(const Op2& a, const Op2& b, f32x4 c) { return mfma(a, b, c); }
  static constexpr int kImages = 2;
  using io_t = float;
  using Op = Op2;
  using Tr = Tr2;
  using Raw = Raw2;
  static __device__ __forceinline__ Raw load_raw(const io_t* p) {
    return Raw2{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)};
  }
  static __device__ __forceinline__ Raw zero_raw() { return Raw2{make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)}; }
  static __device__ __forceinline__ Op op_of(const Raw& r) {
    const float x[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
    Op2 o;
    split8(x, o.h, o.l);
    return o;
  }
  static __device__ __forceinline__ Op load_op(const io_t* p) { return op_of(load_raw(p)); }
  static __device__ __forceinline__ Op zero_op() { return Op2{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}; }
  static __device__ __forceinline__ void stage(uint16_t* img, int lo_off, int off, const Raw& r) {
    const Op2 o = op_of(r);
    *reinterpret_cast<uint4*>(img + off) = o.h;
    *reinterpret_cast<uint4*>(img + lo_off + off) = o.l;
  }
  static __device__ __forceinline__ Op tile_op(const uint16_t* img, int lo_off, int tile, int off) {
    return Op2{tile_frag(img, tile, off), tile_frag(img + lo_off, tile, off)};
  }
  static __device__ __forceinline__ Tr tile_tr(const uint16_t* img, int lo_off, int tile, int off) {
    return Tr2{tile_frag_tr(img, tile, off), tile_frag_tr(img + lo_off, tile, off)};
  }
  static __device__ __forceinline__ Tr zero_tr() { return Tr2{make_uint2(0, 0), make_uint2(0, 0)}; }
  static __device__ __forceinline__ Op join(const Tr& lo, const Tr& hi) {
    return Op2{make_uint4(lo.h.x, lo.h.y, hi.h.x, hi.h.y), make_uint4(lo.l.x, lo.l.y, hi.l.x, hi.l.y)};
  }
  template <typename A>
  static __device__ __forceinline__ Op pack(const A& a, const A& b) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    Op2 o;
    split8(x, o.h, o.l);
    return o;
  }
  template <typename A>
  static __device__ __forceinline__ Op pack_lo(const A& a) {
    const float x[8] = {a[0], a[1], a[2], a[3], 0.f, 0.f, 0.f, 0.f};
    Op2 o;
    split8(x, o.h, o.l);
    return o;
  }
  static __device__ __forceinline__ void to_f32(const Op& o, float (&v)[8]) {
    float hv[8], lv[8];
    Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&o.h), hv);
    Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(&o.l), lv);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = hv[i] + lv[i];
  }
  static __device__ __forceinline__ float to_f32_1(io_t v) { return v; }
  static __device__ __forceinline__ io_t from_f32(float v) { return v; }
  static __device__ __forceinline__ void store4(io_t* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
  }
};

// Cooperative staging of two row sets A and B (`nrows` rows of 64 channels each) into swizzled row-major images
// (PrecSplit: hi and lo image each, the lo image lo_off elements behind);
// rows in [nrows, rows_pad) are zero-filled. NT threads, 8 lanes per row, MAXP >= ceil(rows_pad / (NT/8))
// passes. Row r of set X lives at pX + r * strideX (elements), except row 0 when p0X != nullptr (the cls token
// in front of a frame's patch rows). One pointer per thread + a constant stride per pass keeps the address
// arithmetic out of the way, and ALL global loads are issued before the first LDS write so that the passes
// overlap in flight instead of paying one HBM latency each.
template <typename P, int NT, int MAXP>
__device__ __forceinline__ void stage_rows2(uint16_t* imgA, const typename P::io_t* pA, size_t strideA,
                                            const typename P::io_t* p0A, uint16_t* imgB, const typename P::io_t* pB,
                                            size_t strideB, const typename P::io_t* p0B, int rows_pad, int nrows,
                                            int tid, int lo_off) {
  using io_t = typename P::io_t;
  constexpr int RPP = NT / 8;
  const int c8 = tid & 7, r_in = tid >> 3;
  const io_t* ra = pA + (size_t)r_in * strideA + c8 * 8;
  const io_t* rb = pB + (size_t)r_in * strideB + c8 * 8;
  typename P::Raw va[MAXP], vb[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int r = p * RPP + r_in;
    const bool first = p == 0 && r_in == 0;
    const io_t* sa = (first && p0A != nullptr) ? p0A + c8 * 8 : ra + (size_t)p * RPP * strideA;
    const io_t* sb = (first && p0B != nullptr) ? p0B + c8 * 8 : rb + (size_t)p * RPP * strideB;
    va[p] = P::zero_raw();
    vb[p] = P::zero_raw();
    if (r < nrows) {
      va[p] = P::load_raw(sa);
      vb[p] = P::load_raw(sb);
    }
  }
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int r = p * RPP + r_in;
    if (r < rows_pad) {
      P::stage(imgA, lo_off, img_off(r, c8), va[p]);
      P::stage(imgB, lo_off, img_off(r, c8), vb[p]);
    }
  }
}

// writes a 16x64 f32 tile held in the MFMA C layout (o[dt][r] = X[row g*4+r][col dt*16+c]) as rows of the tensor's
// element type: row i of the tile goes to dst(i) (64 contiguous elements) if valid(i). bf16: through the per-wave LDS
// scratch `ot` ([16][OS] bf16) into 16-byte row pieces; float32 (PrecSplit): the same transposition with the scratch
// read as [16][OS/2] floats is not needed -- a lane holds 4 rows x 1 column per dt, so it goes through `ot` as f32 in
// two halves of 8 rows.
template <typename P, typename DstFn, typename ValidFn>
__device__ __forceinline__ void store_tile_rows(uint16_t* ot, const f32x4 (&o)[4], float mul, int lane, DstFn dst,
                                                ValidFn valid) {
  const int c = lane & 15, g = lane >> 4;
  if constexpr (!P::kSplit) {
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
  } else {
    // the [16][OS] bf16 scratch holds 8 rows of 64 floats (stride OS floats = 288 B): rows g*4+r with (g*4+r) & 8 == hh
    float* of = reinterpret_cast<float*>(ot);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if ((g >> 1) == hh) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) of[((g & 1) * 4 + r) * OS + dt * 16 + c] = o[dt][r] * mul;
      }
      // same wave wrote and reads (LDS operations of one wave complete in order)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int lr = (lane >> 4) + 4 * k, ch = lane & 15;          // 16 lanes x 4 floats per row
        const float4 v = *reinterpret_cast<const float4*>(of + lr * OS + ch * 4);
        const int row = hh * 8 + lr;
        if (valid(row)) *reinterpret_cast<float4*>(dst(row) + ch * 4) = v;
      }
    }
  }
}

}  // namespace attn_mfma
