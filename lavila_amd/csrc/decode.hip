// Narrator decoder (gated-cross-attention GPT-2, lavila/models/gpt2_gated.py) -- the row-wise kernels around the GEMMs,
// gfx950. The reference decodes by re-running the WHOLE prefix through the decoder for every new token
// (narrator.py:118-143, use_cache=False); here a caption advances by ONE row per sequence against a key/value cache, and
// the position lives in DEVICE memory so that one captured hipGraph serves every step of every caption:
//   lvl_gpt2_embed             x[r] = wte[ids[r]] + wpe[pos0 + r % L]                       (gpt2_gated.py:892-895)
//   lvl_gated_add_layernorm    s = res + gate * y;  h = LayerNorm(s)       (the residual adds of GPT2Block.forward,
//                              gpt2_gated.py:442-458,475,483-487, fused with the LayerNorm that reads the sum next)
//   lvl_act_inplace            gelu_new / relu^2                                             (gpt2_gated.py:363-396)
//   lvl_decode_self_attn       append this step's k | v to the cache, attend the new query to rows 0..pos
//                              (gpt2_gated.py:206-238 with layer_past, for query_length 1)
// All HBM-bound row passes: f32 arithmetic, bf16 or f32 storage, 16-byte accesses.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// token + position embedding
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void gpt2_embed_kernel(const int64_t* __restrict__ ids, const T* __restrict__ wte,
                                                         const T* __restrict__ wpe, const int* __restrict__ pos_dev,
                                                         T* __restrict__ out, int L, int D, int vocab, int positions) {
  const int r = blockIdx.x;
  int64_t id = ids[r];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // no wild reads; GPT2LMHeadModel.forward range-checks on the host,
                                                             // decode steps feed the sampler's own tokens
  int p = (pos_dev ? *pos_dev : 0) + r % L;
  p = p >= positions ? positions - 1 : p;
  const T* a = wte + id * D;
  const T* b = wpe + (int64_t)p * D;
  T* o = out + (int64_t)r * D;
  for (int c = threadIdx.x * 8; c < D; c += 128 * 8) {
    float x[8], y[8];
    Elem<T>::load8(a + c, x);
    Elem<T>::load8(b + c, y);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] += y[k];
    Elem<T>::store8(o + c, x);
  }
}

// ------------------------------------------------------------------------------------------------------------
// s = res + gate * y ; h = LN(s)
// ------------------------------------------------------------------------------------------------------------
constexpr int LN_THREADS = 128;
constexpr int LN_CHUNKS = 4;      // D <= 128 * 8 * 4 = 4096 (GPT-2 XL: 1600)

template <typename T>
__global__ __launch_bounds__(LN_THREADS) void gated_add_ln_kernel(const T* res, const T* __restrict__ y,
                                                                  const float* __restrict__ gate,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps,
                                                                  T* sum_out, T* __restrict__ h_out, int D) {
  __shared__ float red[2][LN_THREADS / LVL_WAVE];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float g = gate ? *gate : 1.f;
  const T* rp = res + (int64_t)r * D;
  const T* yp = y ? y + (int64_t)r * D : nullptr;
  float v[LN_CHUNKS][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_CHUNKS; ++i) {
    const int c = (i * LN_THREADS + tid) * 8;
    if (c < D) {
      Elem<T>::load8(rp + c, v[i]);
      if (yp) {
        float t[8];
        Elem<T>::load8(yp + c, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[i][k] = Elem<T>::round(fmaf(g, t[k], v[i][k]));   // the stored residual is what LN sees
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[i][k];
    }
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[0][tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0][0] + red[0][1]) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_CHUNKS; ++i) {
    const int c = (i * LN_THREADS + tid) * 8;
    if (c < D) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = v[i][k] - mean;
        q = fmaf(d, d, q);
      }
    }
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[1][tid >> 6] = q;
  __syncthreads();
  const float rstd = rsqrtf((red[1][0] + red[1][1]) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < LN_CHUNKS; ++i) {
    const int c = (i * LN_THREADS + tid) * 8;
    if (c < D) {
      if (sum_out) Elem<T>::store8(sum_out + (int64_t)r * D + c, v[i]);
      float ga[8], be[8], h[8];
      load8_f32(gamma + c, ga);
      load8_f32(beta + c, be);
#pragma unroll
      for (int k = 0; k < 8; ++k) h[k] = fmaf((v[i][k] - mean) * rstd, ga[k], be[k]);
      Elem<T>::store8(h_out + (int64_t)r * D + c, h);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// activations of the two MLPs
// ------------------------------------------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(256) void act_inplace_kernel(T* __restrict__ u, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float x[8];
    Elem<T>::load8(u + i * 8, x);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (ACT == LVL_ACT_GELU_NEW) {
        const float z = 0.7978845608028654f * (x[k] + 0.044715f * x[k] * x[k] * x[k]);
        const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * z) + 1.f);      // tanh(z), see gemm_skinny.hip
        x[k] = 0.5f * x[k] * (1.f + t);
      } else {
        const float t = fmaxf(x[k], 0.f);
        x[k] = t * t;
      }
    }
    Elem<T>::store8(u + i * 8, x);
  }
}

// ------------------------------------------------------------------------------------------------------------
// one decode step of the causal self-attention: append, then attend
// ------------------------------------------------------------------------------------------------------------
constexpr int SLOTS = 32;

template <typename T>
__global__ __launch_bounds__(256) void decode_self_attn_kernel(const T* __restrict__ qkv, T* __restrict__ cache,
                                                               const int* __restrict__ pos_dev, T* __restrict__ out,
                                                               int Tcap, int H) {
  __shared__ float sm[SLOTS], sl[SLOTS], sacc[SLOTS][64];
  const int tid = threadIdx.x, sub = tid & 7, slot = tid >> 3;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const int D = H * 64;
  int pos = *pos_dev;
  pos = pos < 0 ? 0 : (pos >= Tcap ? Tcap - 1 : pos);       // the host sizes the cache for the caption length
  const T* row = qkv + (int64_t)b * 3 * D + h * 64 + sub * 8;
  float qv[8], kn[8], vn[8];
  Elem<T>::load8(row, qv);
  Elem<T>::load8(row + D, kn);
  Elem<T>::load8(row + 2 * D, vn);
#pragma unroll
  for (int c = 0; c < 8; ++c) qv[c] *= 0.125f;
  T* cb = cache + (int64_t)b * Tcap * 2 * D + h * 64 + sub * 8;
  if (slot == 0) {                                           // this step's key / value join the cache
    Elem<T>::store8(cb + (int64_t)pos * 2 * D, kn);
    Elem<T>::store8(cb + (int64_t)pos * 2 * D + D, vn);
  }
  float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = slot; j <= pos; j += SLOTS) {
    float kx[8], vx[8];
    if (j == pos) {                                          // the new row comes from registers, not back from memory
#pragma unroll
      for (int c = 0; c < 8; ++c) { kx[c] = kn[c]; vx[c] = vn[c]; }
    } else {
      Elem<T>::load8(cb + (int64_t)j * 2 * D, kx);
      Elem<T>::load8(cb + (int64_t)j * 2 * D + D, vx);
    }
    float s = qv[0] * kx[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) s = fmaf(qv[c], kx[c], s);
    s += dpp_move<0xB1>(s);
    s += dpp_move<0x4E>(s);
    s += dpp_move<0x141>(s);
    const float mn = fmaxf(m, s);
    const float corr = __expf(m - mn), p = __expf(s - mn);
    l = l * corr + p;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = fmaf(p, vx[c], acc[c] * corr);
    m = mn;
  }
  if (sub == 0) { sm[slot] = m; sl[slot] = l; }
#pragma unroll
  for (int c = 0; c < 8; ++c) sacc[slot][sub * 8 + c] = acc[c];
  __syncthreads();
  if (tid < 64) {
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) M = fmaxf(M, sm[s]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const float w = sm[s] == -INFINITY ? 0.f : __expf(sm[s] - M);
      L = fmaf(sl[s], w, L);
      o = fmaf(sacc[s][tid], w, o);
    }
    Elem<T>::store(out + (int64_t)b * D + h * 64 + tid, o / L);
  }
}

}  // namespace

extern "C" int lvl_gpt2_embed(const int64_t* ids, const void* wte, const void* wpe, const int* pos_dev, void* out,
                              int rows, int L, int D, int vocab, int positions, int dtype, void* stream) {
  LVL_REQUIRE(rows == 0 || (ids && wte && wpe && out), "gpt2_embed: null pointer");
  LVL_REQUIRE(rows >= 0 && L > 0 && D > 0 && D % 8 == 0 && vocab > 0 && positions > 0,
              "gpt2_embed: bad shape rows=%d L=%d D=%d vocab=%d positions=%d", rows, L, D, vocab, positions);
  LVL_REQUIRE(lvl_aligned16(wte) && lvl_aligned16(wpe) && lvl_aligned16(out), "gpt2_embed: pointers must be 16-byte aligned");
  if (rows == 0) return LVL_OK;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gpt2_embed_kernel<T>), dim3((unsigned)rows), dim3(128), 0,
                                               (hipStream_t)stream, ids, (const T*)wte, (const T*)wpe, pos_dev,
                                               (T*)out, L, D, vocab, positions));
  LVL_CHECK_LAUNCH("gpt2_embed");
  return LVL_OK;
}

extern "C" int lvl_gated_add_layernorm(const void* res, const void* y, const float* gate, const float* gamma,
                                       const float* beta, float eps, void* sum_out, void* h_out, int rows, int D,
                                       int dtype, void* stream) {
  LVL_REQUIRE(rows == 0 || (res && gamma && beta && h_out), "gated_add_layernorm: null pointer");
  LVL_REQUIRE(rows >= 0 && D > 0 && D % 8 == 0 && D <= LN_THREADS * 8 * LN_CHUNKS,
              "gated_add_layernorm: width %d must be a multiple of 8, at most %d", D, LN_THREADS * 8 * LN_CHUNKS);
  LVL_REQUIRE(lvl_aligned16(res) && lvl_aligned16(y) && lvl_aligned16(sum_out) && lvl_aligned16(h_out) &&
                  lvl_aligned16(gamma) && lvl_aligned16(beta), "gated_add_layernorm: pointers must be 16-byte aligned");
  if (rows == 0) return LVL_OK;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gated_add_ln_kernel<T>), dim3((unsigned)rows), dim3(LN_THREADS), 0,
                                               (hipStream_t)stream, (const T*)res, (const T*)y, gate, gamma, beta, eps,
                                               (T*)sum_out, (T*)h_out, D));
  LVL_CHECK_LAUNCH("gated_add_layernorm");
  return LVL_OK;
}

extern "C" int lvl_act_inplace(void* u, int64_t n, int act, int dtype, void* stream) {
  LVL_REQUIRE(n == 0 || u, "act_inplace: null pointer");
  LVL_REQUIRE(n >= 0 && n % 8 == 0 && lvl_aligned16(u), "act_inplace: n %% 8 == 0 and a 16-byte aligned pointer are required");
  LVL_REQUIRE(act == LVL_ACT_GELU_NEW || act == LVL_ACT_SQRELU, "act_inplace: unknown activation %d", act);
  if (n == 0) return LVL_OK;
  const int64_t n8 = n / 8;
  const unsigned grid = (unsigned)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
  if (act == LVL_ACT_GELU_NEW) {
    LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((act_inplace_kernel<T, LVL_ACT_GELU_NEW>), dim3(grid), dim3(256), 0,
                                                 (hipStream_t)stream, (T*)u, n8));
  } else {
    LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((act_inplace_kernel<T, LVL_ACT_SQRELU>), dim3(grid), dim3(256), 0,
                                                 (hipStream_t)stream, (T*)u, n8));
  }
  LVL_CHECK_LAUNCH("act_inplace");
  return LVL_OK;
}

extern "C" int lvl_decode_self_attn(const void* qkv, void* cache, const int* pos_dev, void* out, int B, int Tcap, int H,
                                    int dtype, void* stream) {
  LVL_REQUIRE(B == 0 || (qkv && cache && pos_dev && out), "decode_self_attn: null pointer");
  LVL_REQUIRE(B >= 0 && Tcap > 0 && H > 0, "decode_self_attn: bad shape B=%d Tcap=%d H=%d", B, Tcap, H);
  LVL_REQUIRE(lvl_aligned16(qkv) && lvl_aligned16(cache) && lvl_aligned16(out),
              "decode_self_attn: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((decode_self_attn_kernel<T>), dim3((unsigned)(B * H)), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)qkv, (T*)cache, pos_dev, (T*)out, Tcap, H));
  LVL_CHECK_LAUNCH("decode_self_attn");
  return LVL_OK;
}
