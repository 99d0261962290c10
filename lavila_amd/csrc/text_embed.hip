// Token + positional embedding of the text tower and its backward (models.py:150-153: `token_embedding(text) +
// positional_embedding`; torch's nn.Embedding backward = embedding_dense_backward).
//
// Why own kernels for 4 MB of traffic: above 3072 token rows torch's dense embedding backward sorts the indices with
// rocPRIM's radix sort, which zeroes its histograms with hipMemsetAsync -- under GraphedTrainStep that is a memset NODE in
// the replayed graph, and memset nodes take their fill pattern from recycled memory on this ROCm build
// (profiles/r05_graph_memset_nodes.txt; the benched shape, 256 captions x 32 positions = 8192 rows, takes that path).
// Here: forward = gather + add + one rounding; backward = a deterministic segmented sum without a sort --
//   stage 1  first[v] = min row with token v (integer atomicMin: order-independent), count[v] (integer atomicAdd);
//   stage 2  one workgroup per token row r: the row that is FIRST of its token adds up the dx rows of all its duplicates (its
//            8 waves take 8 contiguous row ranges, each in ascending order; the partial sums are combined in range order:
//            wave-wide compare + ballot over the token list, so a token that occurs once costs one row copy) and stores
//            d table[v] -- no float atomics, run-to-run identical; rows of unused tokens are zeroed by the same launch
//            sequence (lvl_zero_f32: a kernel);
//   d pos[l] = sum over the batch of dx[b, l, :] in batch order (rows >= L zero).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(128) void text_embed_fwd_kernel(const int64_t* __restrict__ tokens, int64_t tok_stride,
                                                             const float* __restrict__ table, const float* __restrict__ pos,
                                                             T* __restrict__ x, int L, int W, int V) {
  const int r = blockIdx.x, b = r / L, l = r - b * L;
  int64_t v = tokens[(int64_t)b * tok_stride + l];
  v = v < 0 ? 0 : (v >= V ? V - 1 : v);                  // never an out-of-range read (nn.Embedding would assert)
  const float* e = table + v * W;
  const float* p = pos + (int64_t)l * W;
  T* o = x + (int64_t)r * W;
  for (int c = threadIdx.x * 4; c < W; c += 128 * 4) {
    float a[4], q[4];
    Elem<float>::load4(e + c, a);
    Elem<float>::load4(p + c, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] += q[i];
    Elem<T>::store4(o + c, a);
  }
}

// first / count initialisation and the census of the token rows (two tiny kernels: the census must see initialised words)
__global__ __launch_bounds__(256) void text_embed_init_kernel(int* __restrict__ first, int* __restrict__ count, int V) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < V) { first[i] = 0x7fffffff; count[i] = 0; }
}

__global__ __launch_bounds__(256) void text_embed_census_kernel(const int64_t* __restrict__ tokens, int64_t tok_stride,
                                                                int* __restrict__ tok32, int* __restrict__ first,
                                                                int* __restrict__ count, int R, int L, int V) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const int b = r / L, l = r - b * L;
  int64_t v = tokens[(int64_t)b * tok_stride + l];
  v = v < 0 ? 0 : (v >= V ? V - 1 : v);
  tok32[r] = (int)v;
  atomicMin(first + v, r);
  atomicAdd(count + v, 1);
}

// d table[v] for the token v of row r, computed by the workgroup of the FIRST row that holds v (the others leave at once).
// A token that occurs once is a row copy. Otherwise the 8 waves cut the rows behind r into 8 contiguous ranges, every wave adds
// up the matching dx rows of its range in ascending order (wave-wide compare + ballot over the token list, two row loads in
// flight), and wave 0 adds the 8 partial sums to the first row in range order -- a fixed order whatever the timing. Heavy
// duplicates are the rule, not the exception: start / end tokens occur once per caption, padding ids thousands of times
// (ragged captions), and their rows would otherwise be one serial chain of dependent loads.
constexpr int kTabWaves = 8;
template <typename T>
__global__ __launch_bounds__(kTabWaves * 64) void text_embed_bwd_table_kernel(const T* __restrict__ dx,
                                                                              const int* __restrict__ tok32,
                                                                              const int* __restrict__ first,
                                                                              const int* __restrict__ count,
                                                                              float* __restrict__ dtable, int R, int W) {
  __shared__ float part[kTabWaves][4][64][8];           // [wave][column group][lane][8 columns] = 64 KiB
  const int r = blockIdx.x;
  const int v = tok32[r];
  if (first[v] != r) return;                             // a later duplicate: its first row adds it up
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = count[v];
  float* out = dtable + (int64_t)v * W;
  float acc[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[k][i] = 0.f;
  auto add_row = [&](int row) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = (k * 64 + lane) * 8;
      if (c < W) {
        float d[8];
        Elem<T>::load8(dx + (int64_t)row * W + c, d);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[k][i] += d[i];
      }
    }
  };
  if (wave == 0) add_row(r);                              // the first row itself: wave 0 starts from it
  if (n > 1) {
    const int span = R - (r + 1);
    const int per = ((span + kTabWaves - 1) / kTabWaves + 63) & ~63;      // rows per wave, a multiple of the scan width
    const int lo = r + 1 + wave * per, hi = min(R, lo + per);
    for (int base = lo; base < hi; base += 64) {
      const int rr = base + lane;
      unsigned long long m = __ballot(rr < hi && tok32[rr] == v);
      while (m) {
        const int j0 = __builtin_ctzll(m);
        m &= m - 1;
        if (m) {                                           // two matches: both rows' loads are issued before either add
          const int j1 = __builtin_ctzll(m);
          m &= m - 1;
          float d0[4][8], d1[4][8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int c = (k * 64 + lane) * 8;
            if (c < W) {
              Elem<T>::load8(dx + (int64_t)(base + j0) * W + c, d0[k]);
              Elem<T>::load8(dx + (int64_t)(base + j1) * W + c, d1[k]);
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if ((k * 64 + lane) * 8 < W) {
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[k][i] = (acc[k][i] + d0[k][i]) + d1[k][i];
            }
        } else {
          add_row(base + j0);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) part[wave][k][lane][i] = acc[k][i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 1; w < kTabWaves; ++w)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[k][i] += part[w][k][lane][i];
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = (k * 64 + lane) * 8;
      if (c < W) Elem<float>::store8(out + c, acc[k]);
    }
  }
}

// d pos[l] = sum_b dx[b, l, :]: 4 batch quarters x 128 column threads per workgroup, the quarters' partial sums added in
// quarter order through LDS (fixed order: deterministic); rows [L, ctx) of dpos are zeroed
template <typename T>
__global__ __launch_bounds__(512) void text_embed_bwd_pos_kernel(const T* __restrict__ dx, float* __restrict__ dpos, int B,
                                                                 int L, int W) {
  __shared__ float part[4][128][4];
  const int l = blockIdx.x, q = threadIdx.x >> 7, ct = threadIdx.x & 127;
  const int b0 = (int)((int64_t)B * q / 4), b1 = (int)((int64_t)B * (q + 1) / 4);
  for (int cb = 0; cb < W; cb += 512) {
    const int c = cb + ct * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (l < L && c < W) {
#pragma unroll 8
      for (int b = b0; b < b1; ++b) {
        float d[4];
        Elem<T>::load4(dx + ((int64_t)b * L + l) * W + c, d);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += d[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) part[q][ct][i] = acc[i];
    __syncthreads();
    if (q == 0 && c < W) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = ((part[0][ct][i] + part[1][ct][i]) + part[2][ct][i]) + part[3][ct][i];
      Elem<float>::store4(dpos + (int64_t)l * W + c, acc);
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int lvl_text_embed_fwd(const int64_t* tokens, int64_t tok_stride, const float* table, const float* pos, void* x,
                                  int B, int L, int W, int V, int dtype, void* stream) {
  LVL_REQUIRE(tokens && table && pos && x, "text_embed_fwd: null pointer");
  LVL_REQUIRE(B >= 0 && L > 0 && V > 0 && W > 0 && W % 4 == 0 && tok_stride >= L,
              "text_embed_fwd: bad shape B=%d L=%d W=%d V=%d stride=%lld", B, L, W, V, (long long)tok_stride);
  LVL_REQUIRE(lvl_aligned16(table) && lvl_aligned16(pos) && (reinterpret_cast<uintptr_t>(x) & 7) == 0,
              "text_embed_fwd: table / pos must be 16-byte aligned, x 8-byte aligned");
  if (B == 0) return LVL_OK;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((text_embed_fwd_kernel<T>), dim3((unsigned)(B * L)), dim3(128), 0,
                                               (hipStream_t)stream, tokens, tok_stride, table, pos, (T*)x, L, W, V));
  LVL_CHECK_LAUNCH("text_embed_fwd");
  return LVL_OK;
}

// int32 words of workspace: tok32 [B L] | first [V] | count [V]
extern "C" int64_t lvl_text_embed_bwd_ws(int B, int L, int V) { return (int64_t)B * L + 2 * (int64_t)V; }

extern "C" int lvl_text_embed_bwd(const void* dx, const int64_t* tokens, int64_t tok_stride, float* dtable, float* dpos,
                                  int* ws, int B, int L, int W, int V, int ctx, int dtype, void* stream) {
  LVL_REQUIRE(dx && tokens && dtable && dpos && ws, "text_embed_bwd: null pointer");
  LVL_REQUIRE(B > 0 && L > 0 && V > 0 && W > 0 && W % 8 == 0 && W <= 2048 && ctx >= L && tok_stride >= L,
              "text_embed_bwd: bad shape B=%d L=%d W=%d V=%d ctx=%d", B, L, W, V, ctx);
  LVL_REQUIRE((int64_t)B * L < (1ll << 31), "text_embed_bwd: too many token rows");
  LVL_REQUIRE(lvl_aligned16(dtable) && lvl_aligned16(dpos) && lvl_aligned16(dx),
              "text_embed_bwd: dtable / dpos / dx must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int R = B * L;
  int* tok32 = ws;
  int* first = ws + R;
  int* count = first + V;
  if (int rc = lvl_zero_f32(dtable, (size_t)V * W, st)) return rc;
  hipLaunchKernelGGL(text_embed_init_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, first, count, V);
  hipLaunchKernelGGL(text_embed_census_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, tokens, tok_stride, tok32,
                     first, count, R, L, V);
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((text_embed_bwd_table_kernel<T>), dim3((unsigned)R), dim3(kTabWaves * 64), 0, st,
                                               (const T*)dx, tok32, first, count, dtable, R, W));
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((text_embed_bwd_pos_kernel<T>), dim3((unsigned)ctx), dim3(512), 0, st,
                                               (const T*)dx, dpos, B, L, W));
  LVL_CHECK_LAUNCH("text_embed_bwd");
  return LVL_OK;
}
