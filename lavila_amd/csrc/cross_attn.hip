// Multi-query cross-attention pooling, gfx950: the core of coca.py's CrossAttention (reference
// lavila/models/coca.py:93-123) as the narrator uses it on top of the video tower (narrator.py:44-49,88-90): NQ learned
// queries x H heads attend to the T tokens of a clip through ONE key/value head of 64 channels shared by all query heads
// (to_kv has 2 * dim_head outputs):
//     out[b, n, h, :] = softmax_j( 0.125 * q[b, n, h, :] . k[b, j, :] ) v[b, j, :]
// Work per clip is tiny next to the tower (0.6 GFLOP vs 185): a latency-tolerant f32 VALU kernel -- 8 lanes per
// (query, head) row with 8 channels each, keys and values staged 64 at a time through LDS, flash-style running
// (max, sum) over blocks of 8 keys, no score tensor.
#include "common.h"

namespace {

constexpr int KC = 64;        // keys per LDS chunk
constexpr int ROWS = 32;      // (query, head) rows per 256-thread workgroup

template <typename T>
__global__ __launch_bounds__(256) void mq_cross_attn_kernel(const T* __restrict__ q, int64_t q_bstride,
                                                            const T* __restrict__ kv, T* __restrict__ out, int NQ,
                                                            int H, int Tk) {
  __shared__ __attribute__((aligned(16))) float ks[KC][64], vs[KC][64];
  const int tid = threadIdx.x, sub = tid & 7, rloc = tid >> 3;
  const int b = blockIdx.y, nrows = NQ * H;
  const int row = blockIdx.x * ROWS + rloc;
  const int rr = row < nrows ? row : nrows - 1;
  float qv[8];
  Elem<T>::load8(q + (int64_t)b * q_bstride + (int64_t)rr * 64 + sub * 8, qv);
#pragma unroll
  for (int c = 0; c < 8; ++c) qv[c] *= 0.125f;
  float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const T* kvb = kv + (int64_t)b * Tk * 128;
  for (int j0 = 0; j0 < Tk; j0 += KC) {
    __syncthreads();
    // stage 64 keys x (k | v): 1024 vectors of 8 elements, 4 per thread
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int vec = p * 256 + tid, key = vec >> 4, c8 = vec & 15;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (j0 + key < Tk) Elem<T>::load8(kvb + (int64_t)(j0 + key) * 128 + c8 * 8, v);
      float* dst = c8 < 8 ? &ks[key][c8 * 8] : &vs[key][(c8 - 8) * 8];
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
    const int nk = Tk - j0 < KC ? Tk - j0 : KC;
    for (int jb = 0; jb < nk; jb += 8) {
      float s[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(&ks[jb + i][sub * 8]);
        const float4 c = *reinterpret_cast<const float4*>(&ks[jb + i][sub * 8 + 4]);
        float d = qv[0] * a.x;
        d = fmaf(qv[1], a.y, d); d = fmaf(qv[2], a.z, d); d = fmaf(qv[3], a.w, d);
        d = fmaf(qv[4], c.x, d); d = fmaf(qv[5], c.y, d); d = fmaf(qv[6], c.z, d); d = fmaf(qv[7], c.w, d);
        d += dpp_move<0xB1>(d);       // lanes of a quad
        d += dpp_move<0x4E>(d);
        d += dpp_move<0x141>(d);      // the other quad of the 8-lane group (mirror within half rows)
        s[i] = jb + i < nk ? d : -INFINITY;
      }
      float mb = s[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) mb = fmaxf(mb, s[i]);
      const float mn = fmaxf(m, mb);
      const float corr = __expf(m - mn);          // exp(-inf) = 0 on the first block
      l *= corr;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] *= corr;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float p = __expf(s[i] - mn);        // masked keys: exp(-inf) = 0
        l += p;
        const float4 a = *reinterpret_cast<const float4*>(&vs[jb + i][sub * 8]);
        const float4 c = *reinterpret_cast<const float4*>(&vs[jb + i][sub * 8 + 4]);
        acc[0] = fmaf(p, a.x, acc[0]); acc[1] = fmaf(p, a.y, acc[1]); acc[2] = fmaf(p, a.z, acc[2]);
        acc[3] = fmaf(p, a.w, acc[3]); acc[4] = fmaf(p, c.x, acc[4]); acc[5] = fmaf(p, c.y, acc[5]);
        acc[6] = fmaf(p, c.z, acc[6]); acc[7] = fmaf(p, c.w, acc[7]);
      }
      m = mn;
    }
  }
  if (row < nrows) {
    const float inv = 1.f / l;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] *= inv;
    Elem<T>::store8(out + ((int64_t)b * nrows + row) * 64 + sub * 8, acc);
  }
}

}  // namespace

extern "C" int lvl_mq_cross_attn_fwd(const void* q, int64_t q_batch_stride, const void* kv, void* out, int B, int NQ,
                                     int H, int Tk, int dtype, void* stream) {
  LVL_REQUIRE(B == 0 || (q && kv && out), "mq_cross_attn_fwd: null pointer");
  LVL_REQUIRE(B >= 0 && NQ > 0 && H > 0 && Tk > 0, "mq_cross_attn_fwd: bad shape B=%d NQ=%d H=%d T=%d", B, NQ, H, Tk);
  LVL_REQUIRE(lvl_aligned16(q) && lvl_aligned16(kv) && lvl_aligned16(out) && q_batch_stride % 8 == 0,
              "mq_cross_attn_fwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  const dim3 grid((unsigned)((NQ * H + ROWS - 1) / ROWS), (unsigned)B);
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((mq_cross_attn_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream,
                                               (const T*)q, q_batch_stride, (const T*)kv, (T*)out, NQ, H, Tk));
  LVL_CHECK_LAUNCH("mq_cross_attn_fwd");
  return LVL_OK;
}
