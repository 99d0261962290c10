// How long does a workgroup of 8 waves need to push one 256x256 bf16 tile (128 KiB) from registers to global memory
// when all 256 CUs do it at the same time? Pattern A = the TN GEMM's epilogue (every store instruction writes 32 rows
// x 32 contiguous bytes), pattern B = full lines (8 rows x 128 bytes per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// RESIDENT: every iteration rewrites the workgroup's FIRST tile (the lines stay in the XCD's L2: what the L2 accepts, not what HBM drains)
template <int PATTERN, bool RESIDENT = false>
__global__ __launch_bounds__(512) void burst(uint4* out, int iters, int N, unsigned long long* clk) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int r5 = lane & 31, hi = lane >> 5;
  uint4 v = make_uint4(tid, lane, wave, 7);
  unsigned long long t = 0;
  for (int it = 0; it < iters; ++it) {
    // tile index: consecutive workgroups take neighbouring column tiles of a [M, N] matrix, then move down
    const int tiles_n = N / 256;
    const long tile = (RESIDENT ? 0L : (long)it * gridDim.x) + blockIdx.x;
    const long tm = tile / tiles_n, tn = tile % tiles_n;
    char* base = (char*)out + (tm * 256 * (long)N + tn * 256) * 2;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < 4; ++j)          // row group of 32 rows
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          long off;
          if (PATTERN == 0) {
            const long m = (j >> 1) * 128 + wm * 64 + (j & 1) * 32 + r5;
            off = (m * N + wn * 64 + i * 32 + 16 * jj + 8 * hi) * 2;
          } else {
            const long m = (j >> 1) * 128 + wm * 64 + (j & 1) * 32 + (i * 2 + jj) * 8 + (lane >> 3);
            off = (m * N + wn * 64 + (lane & 7) * 8) * 2;
          }
          v.x += j;
          *reinterpret_cast<uint4*>(base + off) = v;
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t += __builtin_readcyclecounter() - t0;
  }
  if (tid == 0) clk[blockIdx.x] = t;
}
int main() {
  const int N = 768, iters = 200;
  const long M = 256L * 256 * iters / (N / 256) + 256;
  uint4* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)M * N * 2); hipMalloc(&clk, 256 * 8);
  for (int p = 0; p < 2; ++p) for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (p == 0) hipLaunchKernelGGL(burst<0>, dim3(256), dim3(512), 0, 0, out, iters, N, clk);
    else hipLaunchKernelGGL(burst<1>, dim3(256), dim3(512), 0, 0, out, iters, N, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("pattern %s: %.3f us per 128-KiB tile per CU (all 256 CUs at once), %.2f TB/s aggregate\n",
                    p == 0 ? "A (32 rows x 32 B per store)" : "B (8 rows x 128 B per store)", ms * 1e3 / iters,
                    256.0 * 131072 * iters / (ms * 1e-3) / 1e12);
  }
  for (int p = 0; p < 2; ++p) for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (p == 0) hipLaunchKernelGGL((burst<0, true>), dim3(256), dim3(512), 0, 0, out, iters, N, clk);
    else hipLaunchKernelGGL((burst<1, true>), dim3(256), dim3(512), 0, 0, out, iters, N, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("L2-resident, pattern %s: %.3f us per 128-KiB tile per CU (all 256 CUs at once), %.2f TB/s aggregate\n",
                    p == 0 ? "A (32 rows x 32 B per store)" : "B (8 rows x 128 B per store)", ms * 1e3 / iters,
                    256.0 * 131072 * iters / (ms * 1e-3) / 1e12);
  }
  // fewer workgroups (block b runs on XCD b % 8): is the L2-resident rate a per-CU or a per-XCD figure?
  for (int grid : {256, 128, 64, 32, 8}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((burst<0, true>), dim3(grid), dim3(512), 0, 0, out, 10, N, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((burst<0, true>), dim3(grid), dim3(512), 0, 0, out, iters, N, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("L2-resident, pattern A, %3d workgroups (%2d per XCD): %.3f us per 128-KiB tile per CU, %.2f TB/s aggregate\n", grid,
           grid / 8, ms * 1e3 / iters, (double)grid * 131072 * iters / (ms * 1e-3) / 1e12);
  }
  return 0;
}
