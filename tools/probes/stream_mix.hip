// Micro-benchmark: what HBM delivers for the READ/WRITE MIXES of the row kernels (LayerNorm forward: 1-2 reads + 1-2
// writes; backward: 3 reads + 1 write; time attention: 4 units in + out) on [200960, 768] bf16 tensors (308 MB each, far
// above the 256 MiB Infinity Cache in total), 16 bytes per lane, no arithmetic. Build: hipcc --offload-arch=gfx950 -O3
// tools/probes/stream_mix.hip -o tools/probes/stream_mix ; run it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NR, int NW>
__global__ __launch_bounds__(256) void mix_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                  const uint4* __restrict__ c, uint4* __restrict__ x,
                                                  uint4* __restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint4 v = a[i];
    if (NR >= 2) { const uint4 w = b[i]; v.x ^= w.x; v.y ^= w.y; v.z ^= w.z; v.w ^= w.w; }
    if (NR >= 3) { const uint4 w = c[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    if (NW >= 1) x[i] = v;
    if (NW >= 2) y[i] = make_uint4(v.y, v.x, v.w, v.z);
    if (NW == 0 && v.x == 0x12345678u) x[0] = v;       // keeps the loads alive
  }
}

template <int NR, int NW>
void run(const char* tag, uint4** buf, size_t n, int blocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<NR, NW>), dim3(blocks), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)(NR + NW) * n * 16;
  printf("%-22s %d reads + %d writes of 308 MB: %.3f ms  %.2f TB/s\n", tag, NR, NW, best, bytes / best / 1e9);
}

int main() {
  const size_t n = (size_t)200960 * 768 * 2 / 16;
  uint4* buf[5];
  for (int i = 0; i < 5; ++i) { (void)hipMalloc(&buf[i], n * 16); (void)hipMemset(buf[i], i + 1, n * 16); }
  for (int blocks : {2048, 8192}) {
    printf("grid %d x 256\n", blocks);
    run<1, 0>("read only", buf, n, blocks);
    run<1, 1>("copy", buf, n, blocks);
    run<2, 1>("add+LN no sum (fwd)", buf, n, blocks);
    run<2, 2>("add+LN keep sum (fwd)", buf, n, blocks);
    run<1, 2>("LN of a stored sum", buf, n, blocks);
    run<3, 1>("LN backward", buf, n, blocks);
  }
  return 0;
}
