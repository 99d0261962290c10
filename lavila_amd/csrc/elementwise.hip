// bias+QuickGELU (fwd/bwd), patch gather (patchify) and token assembly, gfx950.
// All HBM-bound single-pass kernels: 16 B per lane, consecutive lanes on consecutive vectors.
#include "common.h"

int lvl_launch_column_reduce(const float* part, int nparts, int width, int seg, float* mid, float* out0, float* out1,
                             float* out2, hipStream_t st);

namespace {

constexpr int kGeluBwdRowBlocks = 1024;   // partial dbias slabs
constexpr int kGeluUnroll = 4;            // rows in flight per thread (all loads issued before the math)

__device__ __forceinline__ float sigmoidf_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ---- a = (u+b) * sigmoid(1.702 (u+b)) -----------------------------------------------------------
// 2-D mapping: thread owns one 8-wide vector column, walks rows with stride gridDim.y, so the bias
// vector is loaded once and (in bwd) the per-column dbias partial stays in registers. kGeluUnroll rows are
// loaded back-to-back before any arithmetic: 4-8 outstanding 16-B loads per lane keep HBM busy.
template <typename T>
__global__ __launch_bounds__(128) void bias_gelu_fwd_kernel(const T* __restrict__ u, const float* __restrict__ bias,
                                                            T* __restrict__ a, int64_t rows, int cols) {
  const int vc = blockIdx.x * blockDim.x + threadIdx.x;
  if (vc * 8 >= cols) return;
  float b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (bias) load8_f32(bias + vc * 8, b);
  const int64_t stride = gridDim.y;
  int64_t r = blockIdx.y;
  for (; r + (kGeluUnroll - 1) * stride < rows; r += kGeluUnroll * stride) {
    float x[kGeluUnroll][8];
#pragma unroll
    for (int k = 0; k < kGeluUnroll; ++k) Elem<T>::load8_nt(u + (r + k * stride) * cols + vc * 8, x[k]);
#pragma unroll
    for (int k = 0; k < kGeluUnroll; ++k) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float y = x[k][j] + b[j];
        o[j] = y * sigmoidf_fast(1.702f * y);
      }
      Elem<T>::store8_nt(a + (r + k * stride) * cols + vc * 8, o);
    }
  }
  for (; r < rows; r += stride) {
    float x[8], o[8];
    Elem<T>::load8_nt(u + r * cols + vc * 8, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float y = x[j] + b[j];
      o[j] = y * sigmoidf_fast(1.702f * y);
    }
    Elem<T>::store8_nt(a + r * cols + vc * 8, o);
  }
}

// du = da * (s + 1.702 y s (1-s)),  s = sigmoid(1.702 y), y = u+b;  dbias = column sums of du
template <typename T>
__global__ __launch_bounds__(128) void bias_gelu_bwd_kernel(const T* __restrict__ da, const T* __restrict__ u,
                                                            const float* __restrict__ bias, T* __restrict__ du,
                                                            float* __restrict__ part, int64_t rows, int cols) {
  const int vc = blockIdx.x * blockDim.x + threadIdx.x;
  if (vc * 8 >= cols) return;
  float b[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (bias) load8_f32(bias + vc * 8, b);
  auto one_row = [&](const float (&x)[8], const float (&g)[8], int64_t r) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float y = x[j] + b[j];
      const float s = sigmoidf_fast(1.702f * y);
      o[j] = g[j] * (s + 1.702f * y * s * (1.0f - s));
      acc[j] += Elem<T>::round(o[j]);      // dbias sums what the weight-grad GEMM will see
    }
    Elem<T>::store8_nt(du + r * cols + vc * 8, o);
  };
  const int64_t stride = gridDim.y;
  int64_t r = blockIdx.y;
  for (; r + (kGeluUnroll - 1) * stride < rows; r += kGeluUnroll * stride) {
    float x[kGeluUnroll][8], g[kGeluUnroll][8];
#pragma unroll
    for (int k = 0; k < kGeluUnroll; ++k) {
      Elem<T>::load8_nt(u + (r + k * stride) * cols + vc * 8, x[k]);
      Elem<T>::load8_nt(da + (r + k * stride) * cols + vc * 8, g[k]);
    }
#pragma unroll
    for (int k = 0; k < kGeluUnroll; ++k) one_row(x[k], g[k], r + k * stride);
  }
  for (; r < rows; r += stride) {
    float x[8], g[8];
    Elem<T>::load8_nt(u + r * cols + vc * 8, x);
    Elem<T>::load8_nt(da + r * cols + vc * 8, g);
    one_row(x, g, r);
  }
  if (part) {
#pragma unroll
    for (int j = 0; j < 8; ++j) part[(size_t)blockIdx.y * cols + vc * 8 + j] = acc[j];
  }
}

// ---- patchify: [B,C,F,H,W] (or [B,F,C,H,W]) f32 -> [B*F*gh*gw, C*P*P] ------------------------------------
// thread = one patch row (P pixels). Thread order follows the INPUT (.., y, px) so reads of a full
// image row are contiguous across the wave; each thread writes P contiguous output elements.
// sc / sf: distance between consecutive channels / frames of one clip in H*W planes (BCFHW: F, 1; BFCHW: 1, C).
template <typename T, int P>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ video, T* __restrict__ out, int B,
                                                       int C, int F, int H, int W, int sc, int sf) {
  const int gw = W / P, gh = H / P;
  const int64_t total = (int64_t)B * C * F * H * gw;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = id;
    const int px = (int)(t % gw); t /= gw;
    const int y = (int)(t % H); t /= H;
    const int f = (int)(t % F); t /= F;
    const int c = (int)(t % C); t /= C;
    const int b = (int)t;
    const int py = y / P, i = y - py * P;
    const float* src = video + ((((int64_t)b * C * F + (int64_t)c * sc + (int64_t)f * sf)) * H + y) * W + (int64_t)px * P;
    const int64_t m = (((int64_t)b * F + f) * gh + py) * gw + px;
    T* dst = out + m * ((int64_t)C * P * P) + ((int64_t)c * P + i) * P;
    if constexpr (P % 8 == 0) {
#pragma unroll
      for (int j = 0; j < P; j += 8) {
        float v[8];
        load8_f32(src + j, v);
        Elem<T>::store8(dst + j, v);
      }
    } else {
#pragma unroll
      for (int j = 0; j < P; j += 2) {   // P even: 8-byte reads (14 px * 4 B = 56 B rows are 8-B aligned)
        const float2 v = *reinterpret_cast<const float2*>(src + j);
        Elem<T>::store(dst + j, v.x);
        Elem<T>::store(dst + j + 1, v.y);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void patchify_generic_kernel(const float* __restrict__ video, T* __restrict__ out,
                                                               int B, int C, int F, int H, int W, int P, int sc,
                                                               int sf) {
  const int gw = W / P, gh = H / P;
  const int64_t total = (int64_t)B * C * F * H * W;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = id;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H); t /= H;
    const int f = (int)(t % F); t /= F;
    const int c = (int)(t % C); t /= C;
    const int b = (int)t;
    const int py = y / P, i = y - py * P, px = x / P, j = x - px * P;
    const int64_t m = (((int64_t)b * F + f) * gh + py) * gw + px;
    Elem<T>::store(out + m * ((int64_t)C * P * P) + ((int64_t)c * P + i) * P + j,
                   video[((((int64_t)b * C * F + (int64_t)c * sc + (int64_t)f * sf)) * H + y) * W + x]);
  }
}

// ---- token assembly --------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const T* __restrict__ pe, const float* __restrict__ cls,
                                                           const float* __restrict__ pos,
                                                           const float* __restrict__ temporal, T* __restrict__ x,
                                                           int B, int F, int N, int D) {
  const int nvec = D >> 3;
  const int T_ = 1 + F * N;
  const int64_t total = (int64_t)B * T_ * nvec;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total;
       id += (int64_t)gridDim.x * blockDim.x) {
    const int vc = (int)(id % nvec);
    const int64_t bt = id / nvec;
    const int t = (int)(bt % T_);
    const int b = (int)(bt / T_);
    float v[8], p[8];
    if (t == 0) {
      load8_f32(cls + vc * 8, v);
      load8_f32(pos + vc * 8, p);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += p[j];
    } else {
      const int f = (t - 1) / N, n = (t - 1) - f * N;
      float q[8];
      Elem<T>::load8(pe + ((int64_t)b * F * N + (t - 1)) * D + vc * 8, v);
      load8_f32(pos + (int64_t)(1 + n) * D + vc * 8, p);
      load8_f32(temporal + (int64_t)f * D + vc * 8, q);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += p[j] + q[j];   // pos+temporal first, as timesformer.py:361
    }
    Elem<T>::store8(x + bt * D + vc * 8, v);
  }
}

// ---- backward of the token assembly: d pos_embed, d temporal_embed, d cls_token from d x ----------------------------------
// (timesformer.py:353-366 under autograd: pos / temporal rows are broadcast over the batch and over frames / locations.)
// Stage 1: P[c][t][:] = sum over the c-th quarter of the batch of dx[b, t, :]  (one pass over dx, f32 partials);
// stage 2: dpos[0] = sum_c P[c][0] (= d cls_token), dpos[1 + n] = sum_{c,f} P[c][1 + f N + n], dtem[f] = sum_{c,n} P[c][1 + f N + n].
constexpr int kEmbedBwdChunks = 4;
template <typename T>
__global__ __launch_bounds__(128) void embed_bwd_stage1_kernel(const T* __restrict__ dx, float* __restrict__ part, int B,
                                                               int T_, int D) {
  const int t = blockIdx.x, c = blockIdx.y;
  const int nvec = D >> 3;
  const int b0 = (int)((int64_t)B * c / kEmbedBwdChunks), b1 = (int)((int64_t)B * (c + 1) / kEmbedBwdChunks);
  for (int vc = threadIdx.x; vc < nvec; vc += blockDim.x) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const T* base = dx + (int64_t)t * D + vc * 8;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
      float x[4][8];
#pragma unroll
      for (int k = 0; k < 4; ++k) Elem<T>::load8(base + (int64_t)(b + k) * T_ * D, x[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += x[k][j];
    }
    for (; b < b1; ++b) {
      float x[8];
      Elem<T>::load8(base + (int64_t)b * T_ * D, x);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += x[j];
    }
    float* dst = part + ((int64_t)c * T_ + t) * D + vc * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = acc[j];
  }
}

// grid (N + 1 + tem_rows, ceil(D / 64)): 64 columns of one output row per workgroup, its 4 waves split the terms of the sum
// (16 for a positional row, 4 N for a temporal row: the first version walked them serially per thread -- 530 us), 8 loads
// in flight per wave, merged through LDS.
__global__ __launch_bounds__(256) void embed_bwd_stage2_kernel(const float* __restrict__ part, float* __restrict__ dpos,
                                                               float* __restrict__ dtem, int F, int N, int D,
                                                               int tem_rows) {
  __shared__ float red[4][64];
  const int T_ = 1 + F * N, r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = blockIdx.y * 64 + lane;
  // term i of row r -> token index t(i); number of terms per batch chunk
  int nterms, t0, tstride;
  if (r == 0) { nterms = 1; t0 = 0; tstride = 0; }                       // cls position
  else if (r <= N) { nterms = F; t0 = r; tstride = N; }                  // positional row r: tokens 1 + f N + (r - 1)
  else { const int f = r - (N + 1); nterms = f < F ? N : 0; t0 = 1 + f * N; tstride = 1; }
  const int total = nterms * kEmbedBwdChunks;
  float acc = 0.f;
  if (d < D) {
    int i = wave;
    for (; i + 7 * 4 < total; i += 8 * 4) {
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = i + k * 4, c = j / nterms, q = j - c * nterms;
        x[k] = part[((int64_t)c * T_ + t0 + (int64_t)q * tstride) * D + d];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += x[k];
    }
    for (; i < total; i += 4) {
      const int c = i / nterms, q = i - c * nterms;
      acc += part[((int64_t)c * T_ + t0 + (int64_t)q * tstride) * D + d];
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && d < D) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (r <= N) dpos[(int64_t)r * D + d] = v;
    else dtem[(int64_t)(r - (N + 1)) * D + d] = v;
  }
}

inline unsigned grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ---- d(bias) of a qkv Linear from its attention core's dq and dO rows (see lvl_qkv_bias_grad) -----------------
// blockIdx.z = 0: the q third of dqkv (row stride 3D), 1: dout (row stride D). Thread = one 8-wide column vector,
// walks rows with stride gridDim.y, 4 rows in flight; partial slab row = [sums of dq | sums of dout].
constexpr int kBiasGradRowBlocks = 1024;
template <typename T>
__global__ __launch_bounds__(128) void qkv_bias_partial_kernel(const T* __restrict__ dqkv, const T* __restrict__ dout,
                                                               float* __restrict__ part, int64_t rows, int D, int src0) {
  const int vc = blockIdx.x * blockDim.x + threadIdx.x;
  if (vc * 8 >= D) return;
  const int src = blockIdx.z + src0;                     // src0 = 1, gridDim.z = 1: dout only; src0 = 0: dq (and dout)
  const T* base = (src == 0 ? dqkv : dout) + vc * 8;
  const int64_t ld = src == 0 ? 3 * (int64_t)D : D;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t stride = gridDim.y;
  int64_t r = blockIdx.y;
  for (; r + 3 * stride < rows; r += 4 * stride) {
    float x[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) Elem<T>::load8_nt(base + (r + k * stride) * ld, x[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += x[k][j];
  }
  for (; r < rows; r += stride) {
    float x[8];
    Elem<T>::load8_nt(base + r * ld, x);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += x[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[(size_t)blockIdx.y * 2 * D + src * D + vc * 8 + j] = acc[j];
}

// ---- weight staging for the Linear layers: f32 master -> bf16 copy and bf16 transposed copy, one pass --------
// 32 x 32 tiles through LDS (33-word rows): the row-major copy feeds the forward GEMM, the transposed copy the
// input-gradient GEMM (both contraction-contiguous). One launch instead of a cast plus a strided transpose copy.
// 64 x 64 tiles, 16-byte loads, 8-byte stores both ways (the 32 x 32 scalar version ran at 0.6 TB/s -- 3 ms per step
// once every weight is re-cast every optimizer step); any N, K (edges masked element-wise).
__device__ __forceinline__ void cast_transpose_tile(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                    uint16_t* __restrict__ dst_t, int N, int K, int n0, int k0,
                                                    float (&tile)[64][65]) {
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;      // 16 column groups of 4 x 16 rows
  const bool vec = (K % 4 == 0) && (N % 4 == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nl = ry + i * 16, n = n0 + nl, k = k0 + cx * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      if (vec && k + 3 < K) {
        const float4 a = *reinterpret_cast<const float4*>(src + (size_t)n * K + k);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        *reinterpret_cast<uint2*>(dst + (size_t)n * K + k) = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < K) {
            v[e] = src[(size_t)n * K + k + e];
            dst[(size_t)n * K + k + e] = f32_to_bf16(v[e]);
          }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[nl][cx * 4 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kl = ry + i * 16, k = k0 + kl, n = n0 + cx * 4;
    if (k < K) {
      const float a = tile[cx * 4][kl], b = tile[cx * 4 + 1][kl], c = tile[cx * 4 + 2][kl], d = tile[cx * 4 + 3][kl];
      if (vec && n + 3 < N) {
        *reinterpret_cast<uint2*>(dst_t + (size_t)k * N + n) = make_uint2(f32x2_to_bf16x2(a, b), f32x2_to_bf16x2(c, d));
      } else {
        const float t[4] = {a, b, c, d};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N) dst_t[(size_t)k * N + n + e] = f32_to_bf16(t[e]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                             uint16_t* __restrict__ dst_t, int N, int K) {
  __shared__ float tile[64][65];
  cast_transpose_tile(src, dst, dst_t, N, K, blockIdx.y * 64, blockIdx.x * 64, tile);
}

// Many weights in ONE launch (round 6: the training step re-casts every Linear weight once per optimizer step -- 121
// launches of ~5 us each, one per weight at its first use; ops.refresh_weight_copies does them at the top of the forward).
// desc[i] = {src, dst, dst_t, N | K << 32, first tile}: workgroup b finds its weight by bisection over the first-tile column.
struct CastDesc { const float* src; uint16_t* dst; uint16_t* dst_t; unsigned N, K; long long tile0; };
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const CastDesc* __restrict__ desc, int count) {
  __shared__ float tile[64][65];
  const long long b = blockIdx.x;
  int lo = 0, hi = count - 1;
  while (lo < hi) {                         // last descriptor whose first tile is <= b
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const CastDesc d = desc[lo];
  const int tiles_k = ((int)d.K + 63) / 64;
  const int t = (int)(b - d.tile0);
  cast_transpose_tile(d.src, d.dst, d.dst_t, (int)d.N, (int)d.K, (t / tiles_k) * 64, (t % tiles_k) * 64, tile);
}

// f32-class operands of the MFMA GEMMs: a float32 matrix as three bf16 TERM images. x = h + l + O(2^-18 |x|) with
// h = bf16(x) (round to nearest even) and l = bf16(x - h) (the subtraction is exact in f32). Image t of element (r, c)
// goes to dst[r * dst_row_stride + t * dst_term_stride + c]:
//   role 0 (the x / dy side of a product):  terms (h, h, l)
//   role 1 (the w / x side):                terms (h, l, h)
// so that contracting image-by-image accumulates h.h' + h.l' + l.h' -- the product to ~2^-17 relative, every partial
// product exact in the f32 accumulator. Along K (dst_term_stride = cols, dst_row_stride = 3 cols) this is the operand of
// lvl_linear_tn's f32-class mode; stacked along the rows (dst_row_stride = cols, dst_term_stride = rows_padded * cols)
// the operand of lvl_linear_wgrad, whose contraction runs over rows.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                           int64_t rows, int cols, int64_t src_row_stride,
                                                           int64_t dst_row_stride, int64_t dst_term_stride, int role) {
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (c4 * 4 >= cols) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const float4 a = *reinterpret_cast<const float4*>(src + r * src_row_stride + c4 * 4);
    const float x[4] = {a.x, a.y, a.z, a.w};
    float l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float h = bf16_to_f32(f32_to_bf16(x[e]));
      l[e] = (fabsf(h) <= 3.38e38f) ? x[e] - h : 0.f;      // inf / nan stay in the h image only
    }
    const uint2 hv = make_uint2(f32x2_to_bf16x2(x[0], x[1]), f32x2_to_bf16x2(x[2], x[3]));
    const uint2 lv = make_uint2(f32x2_to_bf16x2(l[0], l[1]), f32x2_to_bf16x2(l[2], l[3]));
    uint16_t* d = dst + r * dst_row_stride + c4 * 4;
    *reinterpret_cast<uint2*>(d) = hv;
    *reinterpret_cast<uint2*>(d + dst_term_stride) = role == 0 ? hv : lv;
    *reinterpret_cast<uint2*>(d + 2 * dst_term_stride) = role == 0 ? lv : hv;
  }
}

}  // namespace

extern "C" int lvl_bias_quickgelu_fwd(const void* u, const float* bias, void* a, int64_t rows, int cols, int dtype,
                                      void* stream) {
  LVL_REQUIRE(u && a, "bias_quickgelu_fwd: null pointer");
  LVL_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0, "bias_quickgelu_fwd: cols=%d must be a multiple of 8", cols);
  LVL_REQUIRE(lvl_aligned16(u) && lvl_aligned16(a) && lvl_aligned16(bias), "bias_quickgelu_fwd: pointers must be 16-byte aligned");
  if (rows == 0) return LVL_OK;
  const int vcols = cols / 8;
  const unsigned gx = (vcols + 127) / 128;
  int64_t gy = rows < 4096 / gx ? rows : 4096 / gx;
  if (gy < 1) gy = 1;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bias_gelu_fwd_kernel<T>), dim3(gx, (unsigned)gy), dim3(128), 0,
                                               (hipStream_t)stream, (const T*)u, bias, (T*)a, rows, cols));
  LVL_CHECK_LAUNCH("bias_quickgelu_fwd");
  return LVL_OK;
}

extern "C" int lvl_bias_quickgelu_bwd(const void* da, const void* u, const float* bias, void* du, float* dbias,
                                      float* ws, int64_t rows, int cols, int dtype, void* stream) {
  LVL_REQUIRE(da && u && du, "bias_quickgelu_bwd: null pointer");
  LVL_REQUIRE(dbias == nullptr || ws != nullptr, "bias_quickgelu_bwd: dbias needs a workspace");
  LVL_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0, "bias_quickgelu_bwd: cols=%d must be a multiple of 8", cols);
  LVL_REQUIRE(lvl_aligned16(da) && lvl_aligned16(u) && lvl_aligned16(du) && lvl_aligned16(bias), "bias_quickgelu_bwd: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int vcols = cols / 8;
  const unsigned gx = (vcols + 127) / 128;
  int64_t gy = rows < kGeluBwdRowBlocks ? rows : kGeluBwdRowBlocks;
  if (gy < 1) gy = 1;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bias_gelu_bwd_kernel<T>), dim3(gx, (unsigned)gy), dim3(128), 0, st,
                                               (const T*)da, (const T*)u, bias, (T*)du, dbias ? ws : nullptr, rows,
                                               cols));
  LVL_CHECK_LAUNCH("bias_quickgelu_bwd");
  if (dbias)
    return lvl_launch_column_reduce(ws, (int)gy, cols, cols, ws + (size_t)kGeluBwdRowBlocks * cols, dbias, nullptr, nullptr,
                                    st);
  return LVL_OK;
}

int lvl_gelu_bwd_row_blocks() { return kGeluBwdRowBlocks; }

extern "C" int lvl_patchify(const float* video, void* patches, int B, int C, int F, int H, int W, int P,
                            int frame_major, int dtype, void* stream) {
  LVL_REQUIRE(video && patches, "patchify: null pointer");
  const int sc = frame_major ? 1 : F, sf = frame_major ? C : 1;
  LVL_REQUIRE(B >= 0 && C > 0 && F > 0 && H > 0 && W > 0 && P > 0 && H % P == 0 && W % P == 0,
              "patchify: bad shape B=%d C=%d F=%d H=%d W=%d P=%d", B, C, F, H, W, P);
  LVL_REQUIRE(lvl_aligned16(video) && lvl_aligned16(patches), "patchify: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows_total = (int64_t)B * C * F * H * (W / P);
  if (P == 16 && W % 4 == 0) {
    LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((patchify_kernel<T, 16>), dim3(grid_for(rows_total, 256)), dim3(256),
                                                 0, st, video, (T*)patches, B, C, F, H, W, sc, sf));
  } else if (P == 14 && W % 2 == 0 && dtype == LVL_BF16) {
    hipLaunchKernelGGL((patchify_kernel<bf16_t, 14>), dim3(grid_for(rows_total, 256)), dim3(256), 0, st, video,
                       (bf16_t*)patches, B, C, F, H, W, sc, sf);
  } else {
    LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((patchify_generic_kernel<T>),
                                                 dim3(grid_for((int64_t)B * C * F * H * W, 256)), dim3(256), 0, st,
                                                 video, (T*)patches, B, C, F, H, W, P, sc, sf));
  }
  LVL_CHECK_LAUNCH("patchify");
  return LVL_OK;
}

extern "C" int lvl_embed_tokens_fwd(const void* pe, const float* cls, const float* pos, const float* temporal,
                                    void* x, int B, int F, int N, int D, int dtype, void* stream) {
  LVL_REQUIRE(pe && cls && pos && temporal && x, "embed_tokens_fwd: null pointer");
  LVL_REQUIRE(B >= 0 && F > 0 && N > 0 && D > 0 && D % 8 == 0, "embed_tokens_fwd: bad shape B=%d F=%d N=%d D=%d", B, F, N, D);
  LVL_REQUIRE(lvl_aligned16(pe) && lvl_aligned16(cls) && lvl_aligned16(pos) && lvl_aligned16(temporal) && lvl_aligned16(x),
              "embed_tokens_fwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  const int64_t total = (int64_t)B * (1 + F * N) * (D / 8);
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((embed_tokens_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)pe, cls, pos, temporal, (T*)x, B, F, N,
                                               D));
  LVL_CHECK_LAUNCH("embed_tokens_fwd");
  return LVL_OK;
}

extern "C" int64_t lvl_embed_tokens_bwd_ws(int F, int N, int D) { return (int64_t)kEmbedBwdChunks * (1 + (int64_t)F * N) * D; }

extern "C" int lvl_embed_tokens_bwd(const void* dx, float* dpos, float* dtem, float* ws, int B, int F, int N, int D,
                                    int tem_rows, int dtype, void* stream) {
  LVL_REQUIRE(dx && dpos && dtem && ws, "embed_tokens_bwd: null pointer");
  LVL_REQUIRE(B > 0 && F > 0 && N > 0 && D > 0 && D % 8 == 0 && tem_rows >= F,
              "embed_tokens_bwd: bad shape B=%d F=%d N=%d D=%d tem_rows=%d", B, F, N, D, tem_rows);
  LVL_REQUIRE(lvl_aligned16(dx) && lvl_aligned16(ws), "embed_tokens_bwd: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int T_ = 1 + F * N;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((embed_bwd_stage1_kernel<T>), dim3((unsigned)T_, kEmbedBwdChunks), dim3(128),
                                               0, st, (const T*)dx, ws, B, T_, D));
  hipLaunchKernelGGL(embed_bwd_stage2_kernel, dim3((unsigned)(N + 1 + tem_rows), (unsigned)((D + 63) / 64)), dim3(256), 0, st,
                     ws, dpos, dtem, F, N, D, tem_rows);
  LVL_CHECK_LAUNCH("embed_tokens_bwd");
  return LVL_OK;
}

extern "C" int lvl_cast_transpose(const float* src, void* dst, void* dst_t, int N, int K, void* stream) {
  LVL_REQUIRE(src && dst && dst_t, "cast_transpose: null pointer");
  LVL_REQUIRE(N > 0 && K > 0, "cast_transpose: bad shape N=%d K=%d", N, K);
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((K + 63) / 64, (N + 63) / 64), dim3(256), 0, (hipStream_t)stream, src,
                     (uint16_t*)dst, (uint16_t*)dst_t, N, K);
  LVL_CHECK_LAUNCH("cast_transpose");
  return LVL_OK;
}

extern "C" int lvl_cast_transpose_multi(const void* desc, int count, int64_t total_tiles, void* stream) {
  LVL_REQUIRE(desc && count > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "cast_transpose_multi: bad arguments");
  LVL_REQUIRE((reinterpret_cast<uintptr_t>(desc) & 7) == 0, "cast_transpose_multi: descriptor table must be 8-byte aligned");
  hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const CastDesc*)desc, count);
  LVL_CHECK_LAUNCH("cast_transpose_multi");
  return LVL_OK;
}

extern "C" int lvl_split_bf16x3(const float* src, void* dst, int64_t rows, int cols, int64_t src_row_stride,
                                int64_t dst_row_stride, int64_t dst_term_stride, int role, void* stream) {
  LVL_REQUIRE(src && dst, "split_bf16x3: null pointer");
  LVL_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0, "split_bf16x3: cols=%d must be a multiple of 4", cols);
  LVL_REQUIRE(role == 0 || role == 1, "split_bf16x3: role %d", role);
  LVL_REQUIRE(src_row_stride % 4 == 0 && dst_row_stride % 4 == 0 && dst_term_stride % 4 == 0 && lvl_aligned16(src) &&
                  (reinterpret_cast<uintptr_t>(dst) & 7) == 0,
              "split_bf16x3: strides must be multiples of 4 elements, src 16-byte and dst 8-byte aligned");
  if (rows == 0) return LVL_OK;
  const unsigned gx = (unsigned)((cols / 4 + 255) / 256);
  const unsigned gy = (unsigned)(rows < 16384 ? rows : 16384);
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, rows,
                     cols, src_row_stride, dst_row_stride, dst_term_stride, role);
  LVL_CHECK_LAUNCH("split_bf16x3");
  return LVL_OK;
}

// out[k] = sum_n v[n] W[n, k], float32: the column sums of a Linear's INPUT gradient from the column sums of its output
// gradient (dx = dy W  =>  sum_rows dx = (sum_rows dy) W) -- N, K of a few hundred to a few thousand, a few microseconds.
// 1024 threads: lane = one column, 16 waves split the rows (8 loads in flight each), merged through LDS.
__global__ __launch_bounds__(1024) void vec_mat_kernel(const float* __restrict__ v, const float* __restrict__ W,
                                                       float* __restrict__ out, int N, int K) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (k < K) {
    int n = wave;
    for (; n + 7 * 16 < N; n += 8 * 16) {
      float w[8], x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { w[i] = W[(size_t)(n + i * 16) * K + k]; x[i] = v[n + i * 16]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = fmaf(x[i], w[i], acc);
    }
    for (; n < N; n += 16) acc = fmaf(v[n], W[(size_t)n * K + k], acc);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && k < K) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][lane];
    out[k] = s;
  }
}

extern "C" int lvl_vec_mat_f32(const float* v, const float* W, float* out, int N, int K, void* stream) {
  LVL_REQUIRE(v && W && out, "vec_mat_f32: null pointer");
  LVL_REQUIRE(N > 0 && K > 0, "vec_mat_f32: bad shape N=%d K=%d", N, K);
  hipLaunchKernelGGL(vec_mat_kernel, dim3((unsigned)((K + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, v, W, out, N, K);
  LVL_CHECK_LAUNCH("vec_mat_f32");
  return LVL_OK;
}

int lvl_qkv_bias_row_blocks() { return kBiasGradRowBlocks; }
int lvl_colsum_mid_rows();
int64_t lvl_qkv_bias_ws_floats(int D) { return (int64_t)(kBiasGradRowBlocks + lvl_colsum_mid_rows()) * 2 * D; }

// Column sums of the q third of dqkv (-> dbias[0, D)) and / or of dout (-> dbias[2D, 3D)); a null source is skipped and its
// third of dbias left untouched. ws: lvl_qkv_bias_ws_floats.
int lvl_launch_column_reduce_tail(const float* part, int nparts, int width, int seg, float* mid, float* out0, float* out1,
                                  float* out2, float* zero_dst, const float* copy_src, float* copy_dst, int tail_n,
                                  hipStream_t st);

// zero_k: also write the exact zeros of the k third (dbias[D, 2D)); v_src: a ready-made v third to copy into dbias[2D, 3D)
// (only meaningful when dout is null) -- both ride on the reduction's second stage.
int lvl_qkv_bias_sources(const void* dqkv, const void* dout, float* dbias, float* ws, int64_t rows, int D, int dtype,
                         bool zero_k, const float* v_src, hipStream_t st) {
  if (!dqkv && !dout) return LVL_OK;
  int64_t gy = rows < kBiasGradRowBlocks ? rows : kBiasGradRowBlocks;
  const int src0 = dqkv ? 0 : 1, nsrc = (dqkv ? 1 : 0) + (dout ? 1 : 0);
  const dim3 grid((unsigned)((D / 8 + 127) / 128), (unsigned)gy, (unsigned)nsrc);
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((qkv_bias_partial_kernel<T>), grid, dim3(128), 0, st, (const T*)dqkv,
                                               (const T*)dout, ws, rows, D, src0));
  LVL_CHECK_LAUNCH("qkv_bias_sources");
  // partial rows are [sums of dq | sums of dout] (2 D wide); a skipped source's half holds stale scratch and goes nowhere
  return lvl_launch_column_reduce_tail(ws, (int)gy, 2 * D, D, ws + (size_t)kBiasGradRowBlocks * 2 * D,
                                       dqkv ? dbias : nullptr, dout ? dbias + 2 * (size_t)D : nullptr, nullptr,
                                       zero_k ? dbias + D : nullptr, v_src, (v_src && !dout) ? dbias + 2 * (size_t)D : nullptr,
                                       D, st);
}

extern "C" int lvl_qkv_bias_grad(const void* dqkv, const void* dout, float* dbias, float* ws, int64_t rows, int D,
                                 int dtype, void* stream) {
  LVL_REQUIRE(dqkv && dout && dbias && ws, "qkv_bias_grad: null pointer");
  LVL_REQUIRE(rows >= 0 && D > 0 && D % 8 == 0, "qkv_bias_grad: D=%d must be a multiple of 8", D);
  LVL_REQUIRE(lvl_aligned16(dqkv) && lvl_aligned16(dout), "qkv_bias_grad: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = lvl_zero_f32(dbias, (size_t)3 * D, st)) return rc;      // the k third is exactly 0 (a kernel, not a memset node)
  if (rows == 0) return LVL_OK;
  int64_t gy = rows < kBiasGradRowBlocks ? rows : kBiasGradRowBlocks;
  const dim3 grid((unsigned)((D / 8 + 127) / 128), (unsigned)gy, 2);
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((qkv_bias_partial_kernel<T>), grid, dim3(128), 0, st, (const T*)dqkv,
                                               (const T*)dout, ws, rows, D, 0));
  LVL_CHECK_LAUNCH("qkv_bias_grad");
  return lvl_launch_column_reduce(ws, (int)gy, 2 * D, D, ws + (size_t)kBiasGradRowBlocks * 2 * D, dbias,
                                  dbias + 2 * (size_t)D, nullptr, st);
}
