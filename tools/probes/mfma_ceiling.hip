// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate of the whole chip as a function of the LDS fragment-read
// density (ds_read_b128 per MFMA) -- no global memory traffic inside the loop. Build: hipcc --offload-arch=gfx950 -O3
// tools/probes/mfma_ceiling.hip -o /tmp/mfma_ceiling ; run: /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// READS = ds_read_b128 per group of NACC MFMAs; NACC = accumulator tiles per wave = MFMAs per phase (8: the GEMM's 8 waves
// x 128x64; 16: a 4-wave layout, one wave per SIMD owning 128x128 = 256 accumulator registers, 8 fragment reads per 16
// MFMAs = 0.5 per MFMA, the same LDS-DMA bytes per flop: 4 fills per wave and phase)
template <int READS, int WAVES, int BAR, int DMA, int NACC>
__global__ __launch_bounds__(64 * WAVES) void mfma_loop(float* out, int iters, unsigned long long* clk, int rnd,
                                                        const char* src) {
  __shared__ __attribute__((aligned(1024))) uint4 lds[4096 + (DMA ? 4096 : 0)];       // 64 KiB (+ 64 KiB DMA landing zone)
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += blockDim.x) {
    if (rnd) {                       // random bf16 values in (-2, 2): random sign, exponent 0x3e..0x3f, random mantissa
      uint32_t x = (i + 1) * 2654435761u + blockIdx.x * 40503u, w[4];
      for (int k = 0; k < 4; ++k) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = x >> 1;
        w[k] = (r & 0x807f807fu) | 0x3e003e00u | ((r >> 8) & 0x01000100u);
      }
      lds[i] = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      lds[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    }
  }
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  uint4 fa[8], fb[4];
#pragma unroll
  for (int k = 0; k < 8; ++k) fa[k] = lds[(tid * 8 + k) & 4095];
#pragma unroll
  for (int k = 0; k < 4; ++k) fb[k] = lds[(tid * 4 + k + 2048) & 4095];
  // conflict-free fragment addresses: lane l reads 16 B at row (l&31), chunk ((l>>5) ^ ((l>>1)&7)) of 128-B rows
  const int base = ((tid >> 6) * 32 + (lane & 31)) * 8 + (((lane >> 5) ^ ((lane >> 1) & 7)) & 7);
  const unsigned long long t0 = __builtin_readcyclecounter();
  const uint32_t dma_lds = (uint32_t)(uintptr_t)(lds + 4096) + (uint32_t)(tid >> 6) * 8192;
  // every workgroup streams through its own 2 MiB window of an L2 / Infinity-Cache sized buffer (64 MiB)
  const uint32_t dma_lane = (uint32_t)lane * 16 + (uint32_t)(tid >> 6) * 1024;
  const char* dma_base = src + (size_t)(blockIdx.x & 31) * (2u << 20);
  for (int it = 0; it < iters; ++it) {
    if (BAR) {
      if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 0; m < NACC; ++m) {
      // 2 fills of 1 KiB per wave per 8 MFMAs with 8 waves = the GEMM's LDS-DMA rate; 4 per 16 MFMAs with 4 waves
      if (DMA && ((m & 3) == 1) && (NACC == 16 || m == 1 || m == 5)) {
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(dma_lds + (uint32_t)(m >> 2) * 2048u);
        const char* b = dma_base + ((size_t)((it * 4 + (m >> 2)) & 255) * 8192);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(dma_lane), "s"(b)
                     : "memory", "m0");
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[m & 3]),
                                                       __builtin_bit_cast(bf16x8, fa[m & 7]), acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (m < READS) {
        fa[(m + 4) & 7] = lds[(base + (it & 7) * 256 + m * 8) & 4095];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int READS, int WAVES, int BAR = 0, int DMA = 0, int NACC = 8>
void run(const char* tag, float* out, unsigned long long* clk, int iters, int rnd, const char* src = nullptr) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<READS, WAVES, BAR, DMA, NACC>), dim3(256), dim3(64 * WAVES), 0, 0, out, iters, clk, rnd, src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double flops = 256.0 * WAVES * iters * NACC * 32768.0;
    if (rep == 2)
      printf("%s %-30s waves/CU %d  reads/%dmfma %d : %8.3f ms  %7.1f TFLOP/s  shader clock %.3f GHz  mfma-util %.1f %%\n", rnd ? "random" : "const ", tag, WAVES,
             NACC, READS, ms, flops / ms / 1e9, c / (ms * 1e6), 100.0 * iters * NACC * 32.0 * (WAVES / 4) / c);
  }
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
  const int iters = 40000;
  char* src;
  hipMalloc(&src, 64u << 20);
  {   // random bf16 payload for the DMA stream too (data-dependent power)
    uint32_t* h = (uint32_t*)malloc(64u << 20);
    uint32_t x = 12345u;
    for (size_t i = 0; i < (64u << 20) / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 1) & 0x807f807fu) | 0x3e003e00u; }
    hipMemcpy(src, h, 64u << 20, hipMemcpyHostToDevice);
    free(h);
  }
  for (int rnd = 0; rnd < 2; ++rnd) {
    run<0, 8>("regs only", out, clk, iters, rnd);
    run<6, 8>("0.75 rd/mfma", out, clk, iters, rnd);
    run<6, 8, 1, 0>("0.75 rd + barrier/8", out, clk, iters, rnd);
    run<6, 8, 0, 1>("0.75 rd + DMA", out, clk, iters, rnd, src);
    run<6, 8, 1, 1>("0.75 rd + barrier + DMA", out, clk, iters, rnd, src);
    run<0, 8, 1, 1>("regs + barrier + DMA", out, clk, iters, rnd, src);
    // the 4-wave layout (one wave per SIMD, 128x128 per wave): half the fragment reads per MFMA, half the barriers
    run<0, 4, 0, 0, 16>("4w regs only", out, clk, iters / 2, rnd);
    run<8, 4, 0, 0, 16>("4w 0.5 rd/mfma", out, clk, iters / 2, rnd);
    run<8, 4, 1, 0, 16>("4w 0.5 rd + barrier/16", out, clk, iters / 2, rnd);
    run<8, 4, 1, 1, 16>("4w 0.5 rd + barrier + DMA", out, clk, iters / 2, rnd, src);
    run<12, 4, 1, 1, 16>("4w 0.75 rd + barrier + DMA", out, clk, iters / 2, rnd, src);
  }
  return 0;
}
