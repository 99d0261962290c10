// Contrastive-logit contraction on the matrix cores, fused with the row statistics of both cross-entropies.
//
// logits_per_image rows = (scale*img_local) @ txt_all^T and logits_per_text rows = (scale*txt_local) @ img_all^T
// (loss.py:78-79,92-93): one workgroup per (16 local rows, direction); its 4 waves split the G columns into
// 16-column tiles (v_mfma_f32_16x16x32_bf16), keep running (max, sum-exp, sum p*logit, argmax, diagonal) per
// row in the C layout and merge them through shuffles + LDS, so the [B, G] logits slab never exists in HBM.
// Accuracy: inputs may be f32; every operand is split into bf16 hi + bf16 lo and the product is taken as
// hi*hi + hi*lo + lo*hi (three MFMAs, f32 accumulate): ~2^-16 relative error per product, i.e. f32-class logits
// (the 1e-3 parity bar holds with two orders of margin) at bf16 matrix-core speed. The contraction is tiny
// (2*B*G*E flop per direction); what matters is that no pass over a [B, G] tensor is ever made.
#include "attn_mfma_common.h"

namespace {

using namespace attn_mfma;

struct HiLo { uint4 hi, lo; };

// 8 f32 values -> bf16 hi and bf16 lo fragments
__device__ __forceinline__ HiLo split8(const float (&v)[8]) {
  HiLo r;
  float h[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { h[i] = bf16_to_f32(f32_to_bf16(v[i])); l[i] = v[i] - h[i]; }
  r.hi = make_uint4(pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3]), pack_bf16x2(h[4], h[5]), pack_bf16x2(h[6], h[7]));
  r.lo = make_uint4(pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3]), pack_bf16x2(l[4], l[5]), pack_bf16x2(l[6], l[7]));
  return r;
}

template <typename T, int EK>      // EK = E / 32 k-steps
__global__ __launch_bounds__(256) void clip_fwd_mfma_kernel(const T* __restrict__ img_all, const T* __restrict__ txt_all,
                                                            const float* __restrict__ scale_p, int B, int G,
                                                            int row0, float* __restrict__ stats,
                                                            int32_t* __restrict__ argmax, float* __restrict__ logits) {
  constexpr int E = EK * 32;
  __shared__ float red[4][16][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int i0 = blockIdx.x * 16, dir = blockIdx.y;
  const T* A = dir == 0 ? img_all : txt_all;
  const T* Bm = dir == 0 ? txt_all : img_all;
  const float scale = *scale_p;

  // A fragments: local row i0 + c, scaled, split hi/lo (kept in registers for the whole column sweep)
  HiLo a[EK];
  {
    const int ia = i0 + c < B ? i0 + c : B - 1;
    const T* ap = A + (size_t)(row0 + ia) * E + g * 8;
#pragma unroll
    for (int ks = 0; ks < EK; ++ks) {
      float v[8];
      Elem<T>::load8(ap + ks * 32, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= scale;
      a[ks] = split8(v);
    }
  }
  float m[4], l[4], ex[4], best[4], dg[4];
  int bi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; l[r] = 0.f; ex[r] = 0.f; best[r] = -INFINITY; dg[r] = 0.f; bi[r] = 0x7fffffff; }

  const int ntiles = (G + 15) / 16;
#pragma unroll 1
  for (int jt = wave; jt < ntiles; jt += 4) {
    const int j = jt * 16 + c;
    const T* bp = Bm + (size_t)(j < G ? j : G - 1) * E + g * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < EK; ++ks) {
      float v[8];
      Elem<T>::load8(bp + ks * 32, v);
      const HiLo b = split8(v);
      acc = mfma(a[ks].lo, b.hi, acc);
      acc = mfma(a[ks].hi, b.lo, acc);
      acc = mfma(a[ks].hi, b.hi, acc);
    }
    // acc[r] = logit[row i0 + g*4 + r][col j]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + g * 4 + r;
      if (j < G && i < B) {
        const float z = acc[r];
        if (logits) logits[((size_t)dir * B + i) * G + j] = z;
        if (j == row0 + i) dg[r] = z;
        if (z > best[r]) { best[r] = z; bi[r] = j; }          // columns ascend per lane: first maximum kept
        const float mn = fmaxf(m[r], z);
        const float al = __expf(m[r] - mn), p = __expf(z - mn);
        l[r] = l[r] * al + p;
        ex[r] = ex[r] * al + p * z;
        m[r] = mn;
      }
    }
  }
  // merge the 16 column lanes of each row, then the 4 waves
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int o = 1; o <= 8; o <<= 1) {
      const float m2 = __shfl_xor(m[r], o, 64), l2 = __shfl_xor(l[r], o, 64), e2 = __shfl_xor(ex[r], o, 64);
      const float b2 = __shfl_xor(best[r], o, 64), d2 = __shfl_xor(dg[r], o, 64);
      const int i2 = __shfl_xor(bi[r], o, 64);
      const float mn = fmaxf(m[r], m2);
      const float s1 = m[r] == -INFINITY ? 0.f : __expf(m[r] - mn), s2 = m2 == -INFINITY ? 0.f : __expf(m2 - mn);
      l[r] = l[r] * s1 + l2 * s2;
      ex[r] = ex[r] * s1 + e2 * s2;
      m[r] = mn;
      if (b2 > best[r] || (b2 == best[r] && i2 < bi[r])) { best[r] = b2; bi[r] = i2; }
      dg[r] += d2;
    }
    if (c == 0) {
      float* q = red[wave][g * 4 + r];
      q[0] = m[r]; q[1] = l[r]; q[2] = ex[r]; q[3] = best[r]; q[4] = __int_as_float(bi[r]); q[5] = dg[r];
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int row = threadIdx.x, i = i0 + row;
    if (i < B) {
      float M = -INFINITY, bb = -INFINITY, dd = 0.f;
      for (int w = 0; w < 4; ++w) { M = fmaxf(M, red[w][row][0]); bb = fmaxf(bb, red[w][row][3]); dd += red[w][row][5]; }
      float ll = 0.f, ee = 0.f;
      int idx = 0x7fffffff;
      for (int w = 0; w < 4; ++w) {
        const float s2 = red[w][row][0] == -INFINITY ? 0.f : __expf(red[w][row][0] - M);
        ll += red[w][row][1] * s2;
        ee += red[w][row][2] * s2;
        if (red[w][row][3] == bb) idx = min(idx, __float_as_int(red[w][row][4]));
      }
      float* st = stats + ((size_t)dir * B + i) * 4;
      st[0] = M + __logf(ll);
      st[1] = dd;
      st[2] = ee / ll;
      st[3] = bb;
      argmax[dir * B + i] = idx;
    }
  }
}

}  // namespace

bool lvl_clip_mfma_supported(int E) { return E == 64 || E == 128 || E == 256 || E == 512; }

int lvl_clip_fwd_mfma(const void* img_all, const void* txt_all, const float* scale, int B, int G, int E, int row0,
                      float* stats, int32_t* argmax, float* logits, int dtype, hipStream_t st) {
  const dim3 grid((unsigned)((B + 15) / 16), 2), block(256);
#define CLIP_MFMA(TT, EK)                                                                                        \
  hipLaunchKernelGGL((clip_fwd_mfma_kernel<TT, EK>), grid, block, 0, st, (const TT*)img_all, (const TT*)txt_all, \
                     scale, B, G, row0, stats, argmax, logits)
  LVL_DISPATCH_DTYPE(dtype, {
    switch (E) {
      case 64: CLIP_MFMA(T, 2); break;
      case 128: CLIP_MFMA(T, 4); break;
      case 256: CLIP_MFMA(T, 8); break;
      case 512: CLIP_MFMA(T, 16); break;
      default: return lvl_fail(LVL_ENOSYS, "clip_fwd_mfma: E=%d", E);
    }
  });
#undef CLIP_MFMA
  LVL_CHECK_LAUNCH("clip_loss_fwd_mfma");
  return LVL_OK;
}
