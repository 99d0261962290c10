// Shared device/host helpers for the gfx950 kernels. CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lavila_hip.h"

#define LVL_WAVE 64

// ---------------------------------------------------------------------------------------------
// host-side status plumbing
// ---------------------------------------------------------------------------------------------
extern thread_local char lvl_err_buf[512];
int lvl_fail(int code, const char* fmt, ...);

#define LVL_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return lvl_fail(LVL_EINVAL, __VA_ARGS__);       \
  } while (0)

#define LVL_CHECK_LAUNCH(name)                                                       \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) return lvl_fail(LVL_EHIP, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline bool lvl_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Zero n floats with a KERNEL (core.hip). Not hipMemsetAsync: a memset node inside a replayed hipGraph is not reliable on
// this ROCm build -- the replay's fill takes its 16-byte pattern from memory that has been handed to somebody else by then
// (found by NaN-poisoning free device memory between two replays: the "zeroed" accumulators came back as {0, NaN, 0, 0}
// repeated; tools/probe_graph_step_poison.py, profiles/r05_graph_memset_nodes.txt).
int lvl_zero_f32(float* p, size_t n, hipStream_t st);

// Kernels that use more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once.
// compute units the persistent kernels size their grids for: the device's, or the limit set by lvl_set_compute_units
int lvl_persistent_cus();

// One call per (kernel instantiation, device), thread-safe, never inside the hot launch path afterwards (and
// therefore harmless under stream capture once warmed up); the limit is set to the whole 160 KiB of a gfx950 CU.
template <auto Kernel>
inline int lvl_allow_lds() {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return lvl_fail(LVL_EHIP, "hipGetDevice failed");
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    const hipError_t e =
        hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return lvl_fail(LVL_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    done.fetch_or(bit, std::memory_order_release);
  }
  return LVL_OK;
}

// ---------------------------------------------------------------------------------------------
// element types. bf16 is carried as raw uint16 bits; arithmetic is always f32.
// ---------------------------------------------------------------------------------------------
struct bf16_t { uint16_t bits; };

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round-to-nearest-even: one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
typedef __attribute__((ext_vector_type(2))) __bf16 lvl_bf16x2;
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
  const lvl_bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

typedef uint32_t lvl_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float round(float v) { return v; }
  // 8 consecutive elements, 16-byte aligned pairs
  static __device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  static __device__ __forceinline__ void load8_nt(const float* p, float (&v)[8]) { load8(p, v); }
  static __device__ __forceinline__ void store8_nt(float* p, const float (&v)[8]) { store8(p, v); }
  static __device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  static __device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(p->bits); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { p->bits = f32_to_bf16(v); }
  static __device__ __forceinline__ float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
  static __device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    uint4 a;
    a.x = f32x2_to_bf16x2(v[0], v[1]);
    a.y = f32x2_to_bf16x2(v[2], v[3]);
    a.z = f32x2_to_bf16x2(v[4], v[5]);
    a.w = f32x2_to_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = a;
  }
  // streaming variants (non-temporal: the tensor is far larger than L2 + Infinity Cache and is touched once)
  static __device__ __forceinline__ void load8_nt(const bf16_t* p, float (&v)[8]) {
    const lvl_u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const lvl_u32x4*>(p));
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store8_nt(bf16_t* p, const float (&v)[8]) {
    lvl_u32x4 a;
    a.x = f32x2_to_bf16x2(v[0], v[1]);
    a.y = f32x2_to_bf16x2(v[2], v[3]);
    a.z = f32x2_to_bf16x2(v[4], v[5]);
    a.w = f32x2_to_bf16x2(v[6], v[7]);
    __builtin_nontemporal_store(a, reinterpret_cast<lvl_u32x4*>(p));
  }
  static __device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
  }
};

// W consecutive elements (W = 4 or 8). `Raw` is the packed register image of one vector: loads can be issued rows
// ahead (software prefetch) at half the register cost of unpacked f32 for bf16.
template <typename T, int W> struct VecIO;
template <typename T> struct VecIO<T, 8> {
  static __device__ __forceinline__ void load(const T* p, float (&v)[8]) { Elem<T>::load8(p, v); }
  static __device__ __forceinline__ void store(T* p, const float (&v)[8]) { Elem<T>::store8(p, v); }
};
template <typename T> struct VecIO<T, 4> {
  static __device__ __forceinline__ void load(const T* p, float (&v)[4]) { Elem<T>::load4(p, v); }
  static __device__ __forceinline__ void store(T* p, const float (&v)[4]) { Elem<T>::store4(p, v); }
};

template <typename T, int W> struct RawVec;
template <int W> struct RawVec<float, W> {
  float w[W];
  __device__ __forceinline__ void load(const float* p) {
#pragma unroll
    for (int k = 0; k < W; k += 4) {
      const float4 a = *reinterpret_cast<const float4*>(p + k);
      w[k] = a.x; w[k + 1] = a.y; w[k + 2] = a.z; w[k + 3] = a.w;
    }
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = 0.f;
  }
  __device__ __forceinline__ void unpack(float (&v)[W]) const {
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = w[k];
  }
  // an (empty) use the compiler has to have the registers loaded for: puts the wait for a pending load HERE
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int k = 0; k < W; ++k) asm volatile("" : "+v"(w[k]));
  }
};
template <> struct RawVec<bf16_t, 4> {
  uint2 w;
  __device__ __forceinline__ void load(const bf16_t* p) { w = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void zero() { w = make_uint2(0, 0); }
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(w.x), "+v"(w.y)); }
  __device__ __forceinline__ void unpack(float (&v)[4]) const {
    v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
    v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
  }
};
template <> struct RawVec<bf16_t, 8> {
  uint4 w;
  __device__ __forceinline__ void load(const bf16_t* p) { w = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void zero() { w = make_uint4(0, 0, 0, 0); }
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w)); }
  __device__ __forceinline__ void unpack(float (&v)[8]) const {
    v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
    v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
    v[4] = __uint_as_float(w.z << 16); v[5] = __uint_as_float(w.z & 0xffff0000u);
    v[6] = __uint_as_float(w.w << 16); v[7] = __uint_as_float(w.w & 0xffff0000u);
  }
};

__device__ __forceinline__ void load8_f32(const float* p, float (&v)[8]) { Elem<float>::load8(p, v); }

// ---------------------------------------------------------------------------------------------
// wave-level (64-lane) reductions; every lane receives the result.
// DPP within the 16-lane rows (quad xor 1, xor 2, half-mirror, mirror: 4 VALU ops, no LDS crossbar), then the four
// row results travel through SGPRs (v_readlane). ~10 issue slots instead of 6 dependent ds_bpermute round trips.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float read_lane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum over each aligned group of 16 lanes (one DPP row); every lane of the group receives it
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  v += dpp_move<0x140>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);    // row_half_mirror
  v += dpp_move<0x140>(v);    // row_mirror
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  return fmaxf(fmaxf(read_lane(v, 0), read_lane(v, 16)), fmaxf(read_lane(v, 32), read_lane(v, 48)));
}

// dispatch on the runtime dtype tag
#define LVL_DISPATCH_DTYPE(dtype, ...)                                   \
  do {                                                                   \
    if ((dtype) == LVL_F32) { using T = float; __VA_ARGS__; }            \
    else if ((dtype) == LVL_BF16) { using T = bf16_t; __VA_ARGS__; }     \
    else return lvl_fail(LVL_EINVAL, "unknown dtype %d", (int)(dtype));  \
  } while (0)
