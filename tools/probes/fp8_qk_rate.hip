// Does an fp8 QK^T pay on gfx950 at head dim 64? (BASELINE configs[3]: "fp8 MFMA QK^T path".)
// Sustained issue rate, registers only, 256 workgroups x 8 waves, of the three instructions a 16x16 score tile can be
// built from, and the time each needs for ONE 16(query) x 16(key) x 64(channel) tile of S = Q K^T:
//   bf16   v_mfma_f32_16x16x32_bf16             2 instructions per tile (K = 32 channels each)
//   fp8    v_mfma_f32_16x16x32_fp8_fp8          2 instructions per tile (K = 32 as well: the non-scaled fp8 MFMA does not
//                                               widen K on gfx950, it only halves the operand registers)
//   mxfp8  v_mfma_scale_f32_16x16x128_f8f6f4    1 instruction per tile, HALF of its K = 128 slots zero-padded (one head
//                                               has 64 channels; the contraction cannot span two heads)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/fp8_qk_rate.hip -o tools/probes/fp8_qk_rate ; run it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int KIND>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, uint32_t seed) {
  const int tid = threadIdx.x;
  uint32_t x = (tid + 1) * 2654435761u + blockIdx.x * 40503u + seed;
  uint32_t w[16];
  for (int k = 0; k < 16; ++k) { x = x * 1664525u + 1013904223u; w[k] = (x >> 1) & 0x3f3f3f3fu; }   // small fp8 / bf16 bit patterns
  f32x4 acc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (KIND == 0) {
        const uint4 a = make_uint4(w[m] | 0x3c003c00u, w[m + 1] | 0x3c003c00u, w[m + 2] | 0x3c003c00u, w[m + 3] | 0x3c003c00u);
        const uint4 b = make_uint4(w[m + 4] | 0x3c003c00u, w[m + 5] | 0x3c003c00u, w[m + 6] | 0x3c003c00u, w[m + 7] | 0x3c003c00u);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m], 0, 0, 0);
      } else if (KIND == 1) {
        const long a = ((long)w[m] << 32) | w[m + 1], b = ((long)w[m + 2] << 32) | w[m + 3];
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc[m], 0, 0, 0);
      } else {
        i32x8 a, b;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = (int)w[(m + k) & 15]; b[k] = (int)w[(m + k + 5) & 15]; }
        acc[m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[m], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int KIND>
double run(const char* tag, float* out, int iters, int k_per_instr, int instr_per_tile) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<KIND>), dim3(256), dim3(512), 0, 0, out, iters, 17u * rep);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double n_instr = 256.0 * 8 * iters * 8;                      // whole chip
  const double ns_per_instr_per_simd = ms * 1e6 / (n_instr / 1024);  // 1024 SIMDs
  const double tflops = n_instr * 2.0 * 16 * 16 * k_per_instr / ms / 1e9;
  const double tiles_per_us = n_instr / instr_per_tile / (ms * 1e3);
  printf("%-44s %7.3f ms  %6.2f ns/instr/SIMD  %7.1f TFLOP/s nominal  -> %8.1f score tiles (16x16, 64 ch)/us on the chip\n",
         tag, ms, ns_per_instr_per_simd, tflops, tiles_per_us);
  return tiles_per_us;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  const double b = run<0>("bf16  v_mfma_f32_16x16x32_bf16     (2/tile)", out, iters, 32, 2);
  const double f = run<1>("fp8   v_mfma_f32_16x16x32_fp8_fp8  (2/tile)", out, iters, 32, 2);
  const double m = run<2>("mxfp8 v_mfma_scale_16x16x128_f8f6f4 (1/tile, K half used)", out, iters, 128, 1);
  printf("QK^T score-tile rate relative to bf16 at head dim 64: fp8 %.2fx, mxfp8 (half-filled K = 128) %.2fx\n", f / b, m / b);
  return 0;
}
