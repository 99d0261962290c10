// LayerNorm forward/backward, optionally fused with the residual(+bias) add that feeds it. gfx950.
//
// HBM-bound: one pass over [rows, cols]. One 64-lane wave owns one row at a time and keeps the
// whole row in registers, so every operand is read exactly once; statistics are the two-pass (mean, then
// centred variance) form in f32, matching F.layer_norm. Consecutive lanes take consecutive W-element vectors.
// Two vector widths: W = 4 when cols is a multiple of 256 (768 -> exactly 3 vectors per lane, no idle lanes,
// 25 % fewer registers than 2 x 8 -> one more wave per SIMD in the backward), W = 8 otherwise.
// Algorithmic bytes per row (E = element size): fwd cols*E*(n_in + n_out), bwd cols*E*(1 + n_in + 1).
#include <stdlib.h>

#include "common.h"

// The general and the exact-width kernels must agree to the bit (activation checkpointing recomputes a forward that may take
// the other instantiation): no implicit contraction in this file, every fused multiply-add is written as one.
#pragma clang fp contract(off)

namespace {

constexpr int kRowsPerBlock = 4;          // 4 waves per 256-thread workgroup
constexpr int kLnBwdParts = 768;          // partial dgamma/dbeta/dxsum slabs (one per workgroup)

template <int W>
__device__ __forceinline__ void load_f32(const float* p, float (&v)[W]) { VecIO<float, W>::load(p, v); }

// s = x (+ x2) (+ bias)
template <typename T, int W>
__device__ __forceinline__ void load_sum(const T* __restrict__ x, const T* __restrict__ x2,
                                         const float* __restrict__ bias, int64_t off, int c0, float (&v)[W]) {
  VecIO<T, W>::load(x + off, v);
  if (x2 != nullptr) {
    float w[W];
    VecIO<T, W>::load(x2 + off, w);
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] += w[j];
  }
  if (bias != nullptr) {
    float bb[W];
    load_f32<W>(bias + c0, bb);
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] += bb[j];
  }
}

// X2: the row is x + x2 (compile time: the plain LayerNorm then carries no registers for the second operand's two rows in
// flight -- 76 -> 64 VGPRs for 768 columns = 8 instead of 6 waves per SIMD, round 6)
template <typename T, int VPL, int W, bool X2>
struct LnFwdRow {
  RawVec<T, W> x[VPL], x2[X2 ? VPL : 1];
  __device__ __forceinline__ void load(const T* __restrict__ px, const T* __restrict__ px2, int64_t row, int cols,
                                       int lane, int nvec) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        x[i].load(px + row * cols + c * W);
        if constexpr (X2) x2[i].load(px2 + row * cols + c * W);
      }
    }
  }
};

// software-pipelined like the backward: the next row's packed operands are requested before this row is reduced
template <typename T, int VPL, int W, bool X2>
__global__ __launch_bounds__(256) void ln_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ x2, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ s_out,
    T* __restrict__ out, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int cols,
    float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = cols / W;
  const float inv_cols = 1.0f / (float)cols;
  const int64_t stride = (int64_t)gridDim.x * kRowsPerBlock;
  int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + wave;
  LnFwdRow<T, VPL, W, X2> cur, nxt;
  if (row < rows) cur.load(x, x2, row, cols, lane, nvec);
  for (; row < rows; row += stride) {
    if (row + stride < rows) nxt.load(x, x2, row + stride, cols, lane, nvec);
    float v[VPL][W];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        cur.x[i].unpack(v[i]);
        if constexpr (X2) {
          float w[W];
          cur.x2[i].unpack(w);
#pragma unroll
          for (int j = 0; j < W; ++j) v[i][j] += w[j];
        }
        if (bias != nullptr) {
          float bb[W];
          load_f32<W>(bias + c * W, bb);
#pragma unroll
          for (int j = 0; j < W; ++j) v[i][j] += bb[j];
        }
        if (s_out != nullptr) {
          // the sum is what downstream residuals read: round it once, then normalise the rounded value
          VecIO<T, W>::store(s_out + row * cols + c * W, v[i]);
#pragma unroll
          for (int j = 0; j < W; ++j) v[i][j] = Elem<T>::round(v[i][j]);
        }
#pragma unroll
        for (int j = 0; j < W; ++j) sum += v[i][j];
      }
    }
    const float mu = wave_sum(sum) * inv_cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      if (lane + i * 64 < nvec) {
#pragma unroll
        for (int j = 0; j < W; ++j) { const float d = v[i][j] - mu; sq = fmaf(d, d, sq); }
      }
    }
    const float rs = rsqrtf(wave_sum(sq) * inv_cols + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        float g[W], b[W], o[W];
        load_f32<W>(gamma + c * W, g);
        load_f32<W>(beta + c * W, b);
#pragma unroll
        for (int j = 0; j < W; ++j) o[j] = fmaf((v[i][j] - mu) * rs, g[j], b[j]);
        VecIO<T, W>::store(out + row * cols + c * W, o);
      }
    }
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
    cur = nxt;
  }
}

// EXACT-WIDTH forward (cols == VPL * 64 * W, no s_out; round 6): the same arithmetic as ln_fwd_kernel with NO conditional
// vector-memory instruction in the row loop. Why it exists: the compiler's s_waitcnt insertion merges its counters
// conservatively over branches, and ln_fwd_kernel's per-chunk `c < nvec` / `row + stride < rows` / nullable-pointer branches
// left the row loop with `s_waitcnt vmcnt(0)` in front of the reductions (the prefetched NEXT row was waited for too) and
// behind each per-row reload of gamma / beta (three dependent L2 round trips per row). Here gamma, beta (and the optional
// bias) live in registers, the prefetch is unconditional (the last rows re-read row `rows - 1`), mean / rstd are stored by every
// lane (one address), and every wait the compiler places is a counted one: the next row's loads and this row's stores stay in
// flight across the reductions.
template <typename T, int VPL, int W, bool X2, bool BIAS>
__global__ __launch_bounds__(256) void ln_fwd_exact_kernel(
    const T* __restrict__ x, const T* __restrict__ x2, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ out, float* __restrict__ mean,
    float* __restrict__ rstd, int64_t rows, float eps) {
  constexpr int cols = VPL * 64 * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_cols = 1.0f / (float)cols;
  const int64_t stride = (int64_t)gridDim.x * kRowsPerBlock;
  int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + wave;
  if (row >= rows) return;
  float g[VPL][W], b[VPL][W], bb[BIAS ? VPL : 1][W];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    load_f32<W>(gamma + (lane + i * 64) * W, g[i]);
    load_f32<W>(beta + (lane + i * 64) * W, b[i]);
    if constexpr (BIAS) load_f32<W>(bias + (lane + i * 64) * W, bb[i]);
  }
  RawVec<T, W> cur[VPL], cur2[X2 ? VPL : 1], nxt[VPL], nxt2[X2 ? VPL : 1];
  auto load_row = [&](int64_t r, RawVec<T, W> (&a)[VPL], RawVec<T, W> (&a2)[X2 ? VPL : 1]) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      a[i].load(x + r * cols + (lane + i * 64) * W);
      if constexpr (X2) a2[i].load(x2 + r * cols + (lane + i * 64) * W);
    }
  };
  load_row(row, cur, cur2);
  // everything requested so far has landed before the loop is entered: the loop header then merges "nothing pending" with
  // the back edge's counted state instead of a conservative vmcnt(0) on every iteration
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    cur[i].pin();
    if constexpr (X2) cur2[i].pin();
#pragma unroll
    for (int j = 0; j < W; ++j) {
      asm volatile("" : "+v"(g[i][j]), "+v"(b[i][j]));
      if constexpr (BIAS) asm volatile("" : "+v"(bb[i][j]));
    }
  }
  for (; row < rows; row += stride) {
    const int64_t rn = row + stride < rows ? row + stride : rows - 1;      // unconditional prefetch
    load_row(rn, nxt, nxt2);
    float v[VPL][W];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      cur[i].unpack(v[i]);
      if constexpr (X2) {
        float w[W];
        cur2[i].unpack(w);
#pragma unroll
        for (int j = 0; j < W; ++j) v[i][j] += w[j];
      }
      if constexpr (BIAS) {
#pragma unroll
        for (int j = 0; j < W; ++j) v[i][j] += bb[i][j];
      }
#pragma unroll
      for (int j = 0; j < W; ++j) sum += v[i][j];
    }
    const float mu = wave_sum(sum) * inv_cols;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
      for (int j = 0; j < W; ++j) { const float d = v[i][j] - mu; sq = fmaf(d, d, sq); }
    const float rs = rsqrtf(wave_sum(sq) * inv_cols + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      float o[W];
#pragma unroll
      for (int j = 0; j < W; ++j) o[j] = fmaf((v[i][j] - mu) * rs, g[i][j], b[i][j]);
      VecIO<T, W>::store(out + row * cols + (lane + i * 64) * W, o);
    }
    mean[row] = mu;          // every lane, one address: no exec-masked (conditional) store in the loop; both non-null here
    rstd[row] = rs;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      cur[i] = nxt[i];
      if constexpr (X2) cur2[i] = nxt2[i];
    }
  }
}

// dx = rstd * (dy*g - mean(dy*g) - shat * mean(dy*g*shat)) (+ dadd);
// per-workgroup partial slabs [3][cols] = {sum dy*shat, sum dy, sum dx}
// The row loop is software-pipelined: the packed operands of the NEXT row (and its mean/rstd) are requested before
// the current row is reduced, so each wave keeps two rows of loads in flight (4 waves/SIMD would otherwise leave
// HBM idle during the two cross-lane reductions).
template <typename T, int VPL, int W>
struct LnBwdRow {
  RawVec<T, W> dy[VPL], x[VPL], x2[VPL], dadd[VPL];
  float mu, rs;
  __device__ __forceinline__ void load(const T* __restrict__ pdy, const T* __restrict__ px,
                                       const T* __restrict__ px2, const T* __restrict__ pdadd,
                                       const float* __restrict__ mean, const float* __restrict__ rstd, int64_t row,
                                       int cols, int lane, int nvec) {
    mu = mean[row];
    rs = rstd[row];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        const int64_t off = row * cols + c * W;
        dy[i].load(pdy + off);
        x[i].load(px + off);
        if (px2 != nullptr) x2[i].load(px2 + off);
        if (pdadd != nullptr) dadd[i].load(pdadd + off);
      }
    }
  }
};

// the tail of both backward kernels: combine the 4 waves of a workgroup through LDS, then one partial slab per workgroup
template <int VPL, int W>
__device__ __forceinline__ void ln_bwd_combine(float* smem, float* __restrict__ part, float (&ag)[VPL][W], float (&ab)[VPL][W],
                                               float (&ax)[VPL][W], int cols, int nvec, int lane, int wave) {
  if (wave > 0) {
    float* dst = smem + (size_t)(wave - 1) * 3 * cols;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
          dst[c * W + j] = ag[i][j];
          dst[cols + c * W + j] = ab[i][j];
          dst[2 * cols + c * W + j] = ax[i][j];
        }
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    float* pg = part + (size_t)blockIdx.x * 3 * cols;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
          float a = ag[i][j], b = ab[i][j], e = ax[i][j];
          for (int w = 0; w < 3; ++w) {
            a += smem[(size_t)w * 3 * cols + c * W + j];
            b += smem[(size_t)w * 3 * cols + cols + c * W + j];
            e += smem[(size_t)w * 3 * cols + 2 * cols + c * W + j];
          }
          pg[c * W + j] = a;
          pg[cols + c * W + j] = b;
          pg[2 * cols + c * W + j] = e;
        }
      }
    }
  }
}

// EXACT-WIDTH backward (cols == VPL * 64 * W; round 6): ln_bwd_kernel's arithmetic with no conditional vector-memory
// instruction in the row loop (see ln_fwd_exact_kernel: the general kernel's loop waits vmcnt(0) -- for the prefetched next
// row as well -- in front of every reduction). X2: the normalised row is x + x2 (+ bias, zeros when there is none);
// DADD / PLAIN as in the general kernel. mean / rstd of the next row are prefetched with it.
template <typename T, int VPL, int W, bool X2, bool DADD, bool PLAIN>
__global__ __launch_bounds__(256, (VPL * W <= 12 ? (sizeof(T) == 2 ? 3 : 2) : 1)) void ln_bwd_exact_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ x2,
    const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, const T* __restrict__ dadd, T* __restrict__ dx, T* __restrict__ dx_plain,
    float* __restrict__ part, int64_t rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [3 waves][3][cols]
  constexpr int cols = VPL * 64 * W, nvec = VPL * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_cols = 1.0f / (float)cols;
  float g[VPL][W], ag[VPL][W], ab[VPL][W], ax[VPL][W], bb[X2 ? VPL : 1][W];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
#pragma unroll
    for (int j = 0; j < W; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; ax[i][j] = 0.f; }
    load_f32<W>(gamma + (lane + i * 64) * W, g[i]);
    if constexpr (X2) {
#pragma unroll
      for (int j = 0; j < W; ++j) bb[i][j] = 0.f;
    }
  }
  if constexpr (X2) {
    if (bias != nullptr) {            // before the loop: a branch here costs nothing
#pragma unroll
      for (int i = 0; i < VPL; ++i) load_f32<W>(bias + (lane + i * 64) * W, bb[i]);
    }
  }
  const int64_t stride = (int64_t)gridDim.x * kRowsPerBlock;
  int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + wave;
  struct Row {
    RawVec<T, W> dy[VPL], x[VPL], x2[X2 ? VPL : 1], dadd[DADD ? VPL : 1];
    float mu, rs;
  } cur, nxt;
  auto load_row = [&](int64_t r, Row& q) {
    q.mu = mean[r];
    q.rs = rstd[r];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int64_t off = r * cols + (lane + i * 64) * W;
      q.dy[i].load(dy + off);
      q.x[i].load(x + off);
      if constexpr (X2) q.x2[i].load(x2 + off);
      if constexpr (DADD) q.dadd[i].load(dadd + off);
    }
  };
  if (row < rows) {
    load_row(row, cur);
    // everything requested so far has landed before the loop is entered (see ln_fwd_exact_kernel)
    asm volatile("" : "+v"(cur.mu), "+v"(cur.rs));
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      cur.dy[i].pin();
      cur.x[i].pin();
      if constexpr (X2) cur.x2[i].pin();
      if constexpr (DADD) cur.dadd[i].pin();
#pragma unroll
      for (int j = 0; j < W; ++j) {
        asm volatile("" : "+v"(g[i][j]));
        if constexpr (X2) asm volatile("" : "+v"(bb[i][j]));
      }
    }
  }
  for (; row < rows; row += stride) {
    const int64_t rn = row + stride < rows ? row + stride : rows - 1;      // unconditional prefetch
    load_row(rn, nxt);
    const float mu = cur.mu, rs = cur.rs;
    auto xhat = [&](int i, float (&xh)[W]) {
      cur.x[i].unpack(xh);
      if constexpr (X2) {
        float w[W];
        cur.x2[i].unpack(w);
#pragma unroll
        for (int j = 0; j < W; ++j) xh[j] = (xh[j] + w[j]) + bb[i][j];      // the general kernel's order of additions
      }
#pragma unroll
      for (int j = 0; j < W; ++j) xh[j] = (xh[j] - mu) * rs;
    };
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      float xh[W], dv[W];
      xhat(i, xh);
      cur.dy[i].unpack(dv);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const float dgj = dv[j] * g[i][j];
        s1 += dgj;
        s2 = fmaf(dgj, xh[j], s2);
        ag[i][j] = fmaf(dv[j], xh[j], ag[i][j]);
        ab[i][j] += dv[j];
      }
    }
    const float c1 = wave_sum(s1) * inv_cols, c2 = wave_sum(s2) * inv_cols;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      float xh[W], dv[W], o[W];
      xhat(i, xh);
      cur.dy[i].unpack(dv);
#pragma unroll
      for (int j = 0; j < W; ++j) o[j] = rs * fmaf(-xh[j], c2, dv[j] * g[i][j] - c1);
      const int64_t off = row * cols + (lane + i * 64) * W;
      if constexpr (PLAIN) {           // the normalisation's own input gradient leaves separately
#pragma unroll
        for (int j = 0; j < W; ++j) ax[i][j] += Elem<T>::round(o[j]);
        VecIO<T, W>::store(dx_plain + off, o);
      }
      if constexpr (DADD) {
        float e[W];
        cur.dadd[i].unpack(e);
#pragma unroll
        for (int j = 0; j < W; ++j) o[j] += e[j];
      }
      if constexpr (!PLAIN) {
#pragma unroll
        for (int j = 0; j < W; ++j) ax[i][j] += Elem<T>::round(o[j]);
      }
      VecIO<T, W>::store(dx + off, o);
    }
    cur = nxt;
  }
  ln_bwd_combine<VPL, W>(smem, part, ag, ab, ax, cols, nvec, lane, wave);
}

template <typename T, int VPL, int W>
__global__ __launch_bounds__(256, (VPL * W <= 12 ? (sizeof(T) == 2 ? 3 : 2) : 1)) void ln_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ x2,
    const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, const T* __restrict__ dadd, T* __restrict__ dx, T* __restrict__ dx_plain,
    float* __restrict__ part, int64_t rows, int cols) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [3 waves][3][cols]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = cols / W;
  const float inv_cols = 1.0f / (float)cols;
  float g[VPL][W], ag[VPL][W], ab[VPL][W], ax[VPL][W];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int j = 0; j < W; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; ax[i][j] = 0.f; g[i][j] = 0.f; }
    if (c < nvec) load_f32<W>(gamma + c * W, g[i]);
  }
  float bb[VPL][W];     // residual-branch bias (recompute variant only)
  if (bias != nullptr) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
#pragma unroll
      for (int j = 0; j < W; ++j) bb[i][j] = 0.f;
      if (c < nvec) load_f32<W>(bias + c * W, bb[i]);
    }
  }
  const int64_t stride = (int64_t)gridDim.x * kRowsPerBlock;
  int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + wave;
  LnBwdRow<T, VPL, W> cur, nxt;
  if (row < rows) cur.load(dy, x, x2, dadd, mean, rstd, row, cols, lane, nvec);
  for (; row < rows; row += stride) {
    if (row + stride < rows) nxt.load(dy, x, x2, dadd, mean, rstd, row + stride, cols, lane, nvec);
    const float mu = cur.mu, rs = cur.rs;
    // normalised input of vector i, rebuilt from the packed operands (cheaper than holding it across the reduction)
    auto xhat = [&](int i, int c, float (&xh)[W]) {
      cur.x[i].unpack(xh);
      if (x2 != nullptr) {
        float w[W];
        cur.x2[i].unpack(w);
#pragma unroll
        for (int j = 0; j < W; ++j) xh[j] += w[j];
      }
      if (bias != nullptr) {
#pragma unroll
        for (int j = 0; j < W; ++j) xh[j] += bb[i][j];
      }
#pragma unroll
      for (int j = 0; j < W; ++j) xh[j] = (xh[j] - mu) * rs;
    };
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        float xh[W], dv[W];
        xhat(i, c, xh);
        cur.dy[i].unpack(dv);
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const float dgj = dv[j] * g[i][j];
          s1 += dgj;
          s2 = fmaf(dgj, xh[j], s2);
          ag[i][j] = fmaf(dv[j], xh[j], ag[i][j]);
          ab[i][j] += dv[j];
        }
      }
    }
    const float c1 = wave_sum(s1) * inv_cols, c2 = wave_sum(s2) * inv_cols;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        float xh[W], dv[W], o[W];
        xhat(i, c, xh);
        cur.dy[i].unpack(dv);
#pragma unroll
        for (int j = 0; j < W; ++j) o[j] = rs * fmaf(-xh[j], c2, dv[j] * g[i][j] - c1);
        if (dx_plain != nullptr) {       // the normalisation's own input gradient leaves separately (see header)
#pragma unroll
          for (int j = 0; j < W; ++j) ax[i][j] += Elem<T>::round(o[j]);
          VecIO<T, W>::store(dx_plain + row * cols + c * W, o);
        }
        if (dadd != nullptr) {
          float e[W];
          cur.dadd[i].unpack(e);
#pragma unroll
          for (int j = 0; j < W; ++j) o[j] += e[j];
        }
        if (dx_plain == nullptr) {
#pragma unroll
          for (int j = 0; j < W; ++j) ax[i][j] += Elem<T>::round(o[j]);
        }
        VecIO<T, W>::store(dx + row * cols + c * W, o);
      }
    }
    cur = nxt;
  }
  ln_bwd_combine<VPL, W>(smem, part, ag, ab, ax, cols, nvec, lane, wave);
}

// Column sums of a partial slab part[nparts][width] (f32), deterministic, in two coalesced stages:
//   stage 1: grid (width/256, kColsumMid): a workgroup owns 256 columns (one 16-byte vector per lane = 1 KiB per row
//            per wave) and one of kColsumMid row chunks; its 4 waves take the chunk's rows round-robin with every load
//            issued before the first add, merge through LDS and write one row of mid[kColsumMid][width];
//   stage 2: 64 columns x 16 row groups per workgroup sum the kColsumMid rows of mid and scatter the result into up to
//            three `seg`-wide outputs.
// The whole slab (7-19 MB) is in flight at once in stage 1: two short launches (~4 + 3 us) instead of one
// latency chain of 8 dependent round trips per thread behind 144 workgroups (50 us, 99 calls per step in round 2).
constexpr int kColsumMid = 64;

__global__ __launch_bounds__(256) void colsum_stage1_kernel(const float* __restrict__ part, int nparts, int width,
                                                            float* __restrict__ mid) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int rpc = (nparts + kColsumMid - 1) / kColsumMid;
  const int r0 = blockIdx.y * rpc, r1 = min(r0 + rpc, nparts);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < width) {
    int p = r0 + wave;
    for (; p + 12 < r1; p += 16) {          // 4 rows in flight per wave
      const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)p * width + c);
      const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(p + 4) * width + c);
      const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(p + 8) * width + c);
      const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(p + 12) * width + c);
      a.x += (v0.x + v1.x) + (v2.x + v3.x);
      a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z);
      a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; p < r1; p += 4) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)p * width + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && c < width) {
    const float4 b = red[1][lane], d = red[2][lane], e = red[3][lane];
    float4 o;
    o.x = (a.x + b.x) + (d.x + e.x);
    o.y = (a.y + b.y) + (d.y + e.y);
    o.z = (a.z + b.z) + (d.z + e.z);
    o.w = (a.w + b.w) + (d.w + e.w);
    *reinterpret_cast<float4*>(mid + (size_t)blockIdx.y * width + c) = o;
  }
}

// Tail (optional): zero_dst[0, tail_n) = 0 and copy_dst[0, tail_n) = copy_src[...] ride along (the k third of a qkv bias
// gradient is exactly 0 and its v third may arrive ready-made: no memset / copy launches beside the reduction).
__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ mid, int width, int seg,
                                                            float* __restrict__ out0, float* __restrict__ out1,
                                                            float* __restrict__ out2, float* __restrict__ zero_dst,
                                                            const float* __restrict__ copy_src,
                                                            float* __restrict__ copy_dst, int tail_n) {
  __shared__ float4 red[16][16];
  const int cg = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cg * 4;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < tail_n; i += gridDim.x * 256) {
    if (zero_dst) zero_dst[i] = 0.f;
    if (copy_dst) copy_dst[i] = copy_src[i];
  }
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < width) {
    float4 v[kColsumMid / 16];
#pragma unroll
    for (int i = 0; i < kColsumMid / 16; ++i) v[i] = *reinterpret_cast<const float4*>(mid + (size_t)(rg + 16 * i) * width + c);
#pragma unroll
    for (int i = 0; i < kColsumMid / 16; ++i) { a.x += v[i].x; a.y += v[i].y; a.z += v[i].z; a.w += v[i].w; }
  }
  red[rg][cg] = a;
  __syncthreads();
  if (rg == 0 && c < width) {
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 b = red[i][cg];
      o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = c + e;
      const int sg = k / seg;
      float* dst = sg == 0 ? out0 : (sg == 1 ? out1 : out2);
      if (dst && k < width) dst[k - sg * seg] = o[e];
    }
  }
}

// (VPL, W) for a row length: W = 4 with exactly cols/256 vectors per lane when possible
#define LN_DISPATCH(cols, CALL)                          \
  do {                                                   \
    if ((cols) % 256 == 0 && (cols) <= 1024) {           \
      switch ((cols) / 256) {                            \
        case 1: CALL(1, 4); break;                       \
        case 2: CALL(2, 4); break;                       \
        case 3: CALL(3, 4); break;                       \
        default: CALL(4, 4); break;                      \
      }                                                  \
    } else if ((cols) / 8 <= 128) CALL(2, 8);            \
    else if ((cols) / 8 <= 256) CALL(4, 8);              \
    else CALL(8, 8);                                     \
  } while (0)

}  // namespace

int lvl_ln_bwd_parts() { return kLnBwdParts; }

int lvl_colsum_mid_rows() { return kColsumMid; }

// out0/out1/out2 receive consecutive `seg`-wide segments of the column sums of part[nparts][width] (width % 4 == 0);
// mid: kColsumMid * width floats of scratch (the callers place it behind their partial slab)
int lvl_launch_column_reduce_tail(const float* part, int nparts, int width, int seg, float* mid, float* out0, float* out1,
                                  float* out2, float* zero_dst, const float* copy_src, float* copy_dst, int tail_n,
                                  hipStream_t st) {
  hipLaunchKernelGGL(colsum_stage1_kernel, dim3((width + 255) / 256, kColsumMid), dim3(256), 0, st, part, nparts, width,
                     mid);
  hipLaunchKernelGGL(colsum_stage2_kernel, dim3((width + 63) / 64), dim3(256), 0, st, mid, width, seg, out0, out1, out2,
                     zero_dst, copy_src, copy_dst, (zero_dst || copy_dst) ? tail_n : 0);
  LVL_CHECK_LAUNCH("column_reduce");
  return LVL_OK;
}

int lvl_launch_column_reduce(const float* part, int nparts, int width, int seg, float* mid, float* out0, float* out1,
                             float* out2, hipStream_t st) {
  return lvl_launch_column_reduce_tail(part, nparts, width, seg, mid, out0, out1, out2, nullptr, nullptr, nullptr, 0, st);
}

extern "C" int lvl_layernorm_fwd(const void* x, const void* x2, const float* xbias, const float* gamma,
                                 const float* beta, void* s_out, void* y, float* mean, float* rstd, int64_t rows,
                                 int cols, float eps, int dtype, void* stream) {
  LVL_REQUIRE(rows == 0 || (x && gamma && beta && y), "layernorm_fwd: null pointer");
  LVL_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 4096,
              "layernorm_fwd: cols=%d must be a multiple of 8, <= 4096", cols);
  LVL_REQUIRE(lvl_aligned16(x) && lvl_aligned16(x2) && lvl_aligned16(xbias) && lvl_aligned16(y) &&
                  lvl_aligned16(s_out) && lvl_aligned16(gamma) && lvl_aligned16(beta),
              "layernorm_fwd: pointers must be 16-byte aligned");
  if (rows == 0) return LVL_OK;
  int64_t blocks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
  // long per-wave row chains: 2 x the resident workgroups at 6 waves/SIMD for the two-operand form (76 VGPRs); the plain form
  // (56 VGPRs, 8 waves/SIMD) measured best with 4 x its resident workgroups (0.136 ms at 8192 against 0.142 at 3072 and 0.176
  // at 2048 for 200 960 rows of 768: profiles/r06_rowops_ln_fwd.txt)
  const int64_t cap = x2 == nullptr ? 8192 : 3072;
  static const bool exact_off = getenv("LAVILA_LN_EXACT") && atoi(getenv("LAVILA_LN_EXACT")) == 0;      // A/B switch
  const bool exact = !exact_off && s_out == nullptr && mean != nullptr && rstd != nullptr;
  if (blocks > cap) blocks = cap;
#define LN_FWD_X(TT, VPL, W, X2)                                                                                \
  hipLaunchKernelGGL((ln_fwd_kernel<TT, VPL, W, X2>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, \
                     (const TT*)x, (const TT*)x2, xbias, gamma, beta, (TT*)s_out, (TT*)y, mean, rstd, rows, cols, eps)
#define LN_FWD_E(TT, VPL, W, X2, BIAS)                                                                             \
  hipLaunchKernelGGL((ln_fwd_exact_kernel<TT, VPL, W, X2, BIAS>), dim3((unsigned)blocks), dim3(256), 0,              \
                     (hipStream_t)stream, (const TT*)x, (const TT*)x2, xbias, gamma, beta, (TT*)y, mean, rstd, rows, eps)
#define LN_FWD_T(TT, VPL, W)                                                              \
  do {                                                                                    \
    if (exact && (VPL) * 64 * (W) == cols && x2 == nullptr && xbias == nullptr) LN_FWD_E(TT, VPL, W, false, false); \
    else if (exact && (VPL) * 64 * (W) == cols && x2 != nullptr && xbias != nullptr) LN_FWD_E(TT, VPL, W, true, true); \
    else if (exact && (VPL) * 64 * (W) == cols && x2 != nullptr) LN_FWD_E(TT, VPL, W, true, false); \
    else if (x2 != nullptr) LN_FWD_X(TT, VPL, W, true);                                   \
    else LN_FWD_X(TT, VPL, W, false);                                                     \
  } while (0)
#define LN_FWD(VPL, W) LN_FWD_T(T, VPL, W)
  LVL_DISPATCH_DTYPE(dtype, LN_DISPATCH(cols, LN_FWD));
#undef LN_FWD
#undef LN_FWD_T
#undef LN_FWD_X
#undef LN_FWD_E
  LVL_CHECK_LAUNCH("layernorm_fwd");
  return LVL_OK;
}

extern "C" int lvl_layernorm_bwd(const void* dy, const void* x, const void* x2, const float* xbias,
                                 const float* gamma, const float* mean, const float* rstd, const void* dadd,
                                 void* dx, void* dx_plain, float* dgamma, float* dbeta, float* dxsum, float* ws,
                                 int64_t rows, int cols, int dtype, void* stream) {
  LVL_REQUIRE((rows == 0 || (dy && x && mean && rstd && dx)) && gamma && ws, "layernorm_bwd: null pointer");
  LVL_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 4096,
              "layernorm_bwd: cols=%d must be a multiple of 8, <= 4096", cols);
  LVL_REQUIRE(lvl_aligned16(dy) && lvl_aligned16(x) && lvl_aligned16(x2) && lvl_aligned16(xbias) &&
                  lvl_aligned16(dadd) && lvl_aligned16(dx) && lvl_aligned16(dx_plain) && lvl_aligned16(gamma),
              "layernorm_bwd: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
  if (blocks > kLnBwdParts) blocks = kLnBwdParts;
  if (blocks < 1) blocks = 1;
  const size_t shmem = (size_t)3 * 3 * cols * sizeof(float);
  static const bool exact_off = getenv("LAVILA_LN_EXACT") && atoi(getenv("LAVILA_LN_EXACT")) == 0;      // A/B switch
  const bool exact = !exact_off;
#define LN_BWD_E(TT, VPL, W, X2, DADD, PLAIN)                                                                      \
  do {                                                                                                            \
    if (shmem > 64 * 1024)                                                                                        \
      if (int rc = lvl_allow_lds<ln_bwd_exact_kernel<TT, VPL, W, X2, DADD, PLAIN>>()) return rc;                  \
    hipLaunchKernelGGL((ln_bwd_exact_kernel<TT, VPL, W, X2, DADD, PLAIN>), dim3((unsigned)blocks), dim3(256), shmem, \
                       st, (const TT*)dy, (const TT*)x, (const TT*)x2, xbias, gamma, mean, rstd, (const TT*)dadd,  \
                       (TT*)dx, (TT*)dx_plain, ws, rows);                                                         \
  } while (0)
#define LN_BWD_G(TT, VPL, W)                                                                                    \
  do {                                                                                                          \
    if (shmem > 64 * 1024)                                                                                      \
      if (int rc = lvl_allow_lds<ln_bwd_kernel<TT, VPL, W>>()) return rc;                                       \
    hipLaunchKernelGGL((ln_bwd_kernel<TT, VPL, W>), dim3((unsigned)blocks), dim3(256), shmem, st, (const TT*)dy, \
                       (const TT*)x, (const TT*)x2, xbias, gamma, mean, rstd, (const TT*)dadd, (TT*)dx,          \
                       (TT*)dx_plain, ws, rows,                                                                   \
                       cols);                                                                                   \
  } while (0)
// the exact-width kernel for the operand combinations of the training step (bf16), the general kernel otherwise
#define LN_BWD_T(TT, VPL, W)                                                                                    \
  do {                                                                                                          \
    constexpr bool kHalf = sizeof(TT) == 2;                                                                     \
    const bool ex = kHalf && exact && (VPL) * 64 * (W) == cols && rows > 0;                                     \
    if (ex && !x2 && !xbias && !dadd && !dx_plain) LN_BWD_E(TT, VPL, W, false, false, false);                   \
    else if (ex && !x2 && !xbias && dadd && !dx_plain) LN_BWD_E(TT, VPL, W, false, true, false);                \
    else if (ex && x2 && !dadd && !dx_plain) LN_BWD_E(TT, VPL, W, true, false, false);                          \
    else if (ex && x2 && dadd && dx_plain) LN_BWD_E(TT, VPL, W, true, true, true);                              \
    else LN_BWD_G(TT, VPL, W);                                                                                  \
  } while (0)
#define LN_BWD(VPL, W) LN_BWD_T(T, VPL, W)
  LVL_DISPATCH_DTYPE(dtype, LN_DISPATCH(cols, LN_BWD));
#undef LN_BWD
#undef LN_BWD_T
#undef LN_BWD_G
#undef LN_BWD_E
  LVL_CHECK_LAUNCH("layernorm_bwd");
  return lvl_launch_column_reduce(ws, (int)blocks, 3 * cols, cols, ws + (size_t)kLnBwdParts * 3 * cols, dgamma, dbeta,
                                  dxsum, st);
}
