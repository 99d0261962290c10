// Probe of ds_read_b64_tr_b16 semantics on gfx950: prints, for every lane, which LDS elements it got.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
  const unsigned short* p = lds + (g * 4 + (m >> 2)) * 80 + (m & 3) * 4;   // row g*4 + m/4, cols (m%4)*4..+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned short *d, *dout;
  hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    int m = l & 15, g = l >> 4;
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      int e = o[l * 4 + j];
      printf(" (r%d,c%d)", e / 80, e % 80);
      if (e != (g * 4 + j) * 80 + m) ok = 0;
    }
    printf("\n");
  }
  printf("HYPOTHESIS out[lane][j] = row (g*4+j), col (lane&15): %s\n", ok ? "CONFIRMED" : "REJECTED");
  return 0;
}
