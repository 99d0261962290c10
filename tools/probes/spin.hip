// Occupies `wgs` compute units for about `cycles` shader cycles each launch: a stand-in for the channel workgroups of a
// collective running beside the step (tools/probe_cu_contention.py). 256 threads, 64 KiB of LDS so that no two of
// them share a CU.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void spin_kernel(long cycles, int* sink) {
  __shared__ int pad[16384];
  pad[threadIdx.x] = threadIdx.x;
  const long t0 = clock64();
  int v = 0;
  while (clock64() - t0 < cycles) v += pad[(threadIdx.x + v) & 16383] & 1;
  if (v == -1) *sink = v;
}
extern "C" int spin_launch(int wgs, long cycles, int* sink, void* stream) {
  hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, cycles, sink);
  return (int)hipGetLastError();
}
