// Micro-benchmark: VALU cost of the QuickGELU epilogue forms of the TN GEMM (128 f32 accumulator values per lane -> bf16),
// 256 workgroups x 8 waves (two waves per SIMD, as in the GEMM), no memory traffic inside the loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_gelu.hip -o /tmp/valu_gelu ; run: /tmp/valu_gelu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ uint32_t pk(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float sigmoid1702(float u) { return __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * u)); }

// FORM 0: round0's scalar code (u rounded, two multiplies in front of exp)
// FORM 1: packed f32 (v_pk_mul / v_pk_add), u rounded
// FORM 2: scalar, one constant multiply, u rounded
// FORM 3: scalar, u NOT rounded, + derivative output
// FORM 4: packed, u NOT rounded, + derivative output
// FORM 5: bare conversion (epilogue 0): pack only
// FORM 6: multiply by an unpacked bf16 aux + pack (epilogue 5 without the column sums)
// FORM 7: old epilogue 2: y * quickgelu'(aux) scalar
template <int FORM>
__global__ __launch_bounds__(512) void k(uint32_t* out, const float* in, int iters) {
  float v[128];
  const int t = blockIdx.x * 512 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 128; ++i) v[i] = in[(t + i * 7) & 1023];
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 128; i += 2) {
      float a = v[i], b = v[i + 1];
      uint32_t o = 0, o2 = 0;
      if (FORM == 0) {
        const uint32_t u = pk(a, b);
        const float ua = __uint_as_float(u << 16), ub = __uint_as_float(u & 0xffff0000u);
        o = pk(ua * sigmoid1702(ua), ub * sigmoid1702(ub));
        o2 = u;
      } else if (FORM == 1 || FORM == 4) {
        f32x2 uu;
        if (FORM == 1) {
          const uint32_t u = pk(a, b);
          uu = f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
          o2 = u;
        } else {
          uu = f32x2{a, b};
        }
        const f32x2 z = uu * (-1.702f * 1.44269504088896341f);
        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + 1.f;
        const f32x2 r = f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        const f32x2 y = uu * r;
        o = pk(y.x, y.y);
        if (FORM == 4) {
          const f32x2 g = __builtin_elementwise_fma((1.f - r) * y, f32x2{1.702f, 1.702f}, r);
          o2 = pk(g.x, g.y);
        }
      } else if (FORM == 2 || FORM == 3) {
        float ua = a, ub = b;
        if (FORM == 2) {
          const uint32_t u = pk(a, b);
          ua = __uint_as_float(u << 16), ub = __uint_as_float(u & 0xffff0000u);
          o2 = u;
        }
        const float c = -1.702f * 1.44269504088896341f;
        const float ra = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(c * ua));
        const float rb = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(c * ub));
        const float ya = ua * ra, yb = ub * rb;
        o = pk(ya, yb);
        if (FORM == 3) o2 = pk(fmaf((1.f - ra) * ya, 1.702f, ra), fmaf((1.f - rb) * yb, 1.702f, rb));
      } else if (FORM == 5) {
        o = pk(a, b);
      } else if (FORM == 6) {
        const uint32_t aux = __float_as_uint(v[(i + 64) & 127]);
        o = pk(a * __uint_as_float(aux << 16), b * __uint_as_float(aux & 0xffff0000u));
      } else if (FORM == 7) {
        const uint32_t aux = __float_as_uint(v[(i + 64) & 127]);
        const float ua = __uint_as_float(aux << 16), ub = __uint_as_float(aux & 0xffff0000u);
        const float sa = sigmoid1702(ua), sb = sigmoid1702(ub);
        o = pk(a * (sa * (1.f + 1.702f * ua * (1.f - sa))), b * (sb * (1.f + 1.702f * ub * (1.f - sb))));
      }
      acc ^= o + o2 * 3u;
      v[i] = a + __uint_as_float((o & 0x007f0000u) | 0x33800000u);      // keeps the chain data-dependent without cost
    }
  }
  out[t] = acc;
}

template <int FORM>
void run(const char* name, uint32_t* out, float* in, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(512), 0, 0, out, in, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(512), 0, 0, out, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f us per 128-element pass (two waves per SIMD)\n", name, ms * 1e3 / iters);
}

int main() {
  uint32_t* out;
  float* in;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&in, 1024 * 4);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 200 - 100) * 0.03f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    run<5>("5 pack only (epilogue 0)", out, in, iters);
    run<0>("0 QuickGELU scalar, round-0 code", out, in, iters);
    run<1>("1 QuickGELU packed f32", out, in, iters);
    run<2>("2 QuickGELU scalar, one constant", out, in, iters);
    run<3>("3 QuickGELU + derivative scalar, u not rounded", out, in, iters);
    run<4>("4 QuickGELU + derivative packed, u not rounded", out, in, iters);
    run<6>("6 multiply by aux (epilogue 5)", out, in, iters);
    run<7>("7 multiply by quickgelu'(aux) scalar (epilogue 2)", out, in, iters);
  }
  return 0;
}
