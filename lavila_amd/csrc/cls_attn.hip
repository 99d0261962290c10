// Attention of ONE query per (sample, head) -- the cls token -- over all T tokens of the clip, gfx950.
// In the LAST block of a cls-pooled forward (SpaceTimeTransformer.forward: norm(x)[:, 0], timesformer.py:377) only the
// cls row of the space attention's output is read, and the cls query attends to every token (timesformer.py:116-119):
//     out[b, h, :] = softmax_j(0.125 * q[b, h, :] . k[b, j, h, :]) v[b, j, h, :],   j = 0 .. T-1
// q: [B, H*64] (projected from the cls rows only), kv: [B, T, 2*H*64] = k | v as a Linear with the k and v thirds of the
// qkv weight writes them. One workgroup per (b, h): 32 key slots x 8 lanes (8 channels each, one 128-byte row per slot),
// one pass over K and V with a flash-style running (max, sum, acc) per slot, merged through LDS. HBM-bound: the kernel
// reads kv once (forward) and reads kv + writes dkv once (backward); f32 arithmetic for both element types.
#include "common.h"

namespace {

constexpr int SLOTS = 32;      // keys in flight per workgroup step (256 threads / 8 lanes)

template <typename T>
__global__ __launch_bounds__(256) void cls_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                           T* __restrict__ out, float* __restrict__ lse, int Tk,
                                                           int H, int qrep) {
  __shared__ float sm[SLOTS], sl[SLOTS], sacc[SLOTS][64];
  const int tid = threadIdx.x, sub = tid & 7, slot = tid >> 3;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const int D = H * 64;
  float qv[8];
  Elem<T>::load8(q + (int64_t)b * D + h * 64 + sub * 8, qv);
#pragma unroll
  for (int c = 0; c < 8; ++c) qv[c] *= 0.125f;
  const T* kb = kv + (int64_t)(b / qrep) * Tk * 2 * D + h * 64 + sub * 8;     // qrep consecutive query rows share a context
  float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = slot; j < Tk; j += SLOTS) {
    float kx[8], vx[8];
    Elem<T>::load8(kb + (int64_t)j * 2 * D, kx);
    Elem<T>::load8(kb + (int64_t)j * 2 * D + D, vx);
    float s = qv[0] * kx[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) s = fmaf(qv[c], kx[c], s);
    s += dpp_move<0xB1>(s);
    s += dpp_move<0x4E>(s);
    s += dpp_move<0x141>(s);           // all 8 lanes of the slot hold the score
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
  if (tid < 64) {       // channel tid: merge the slots
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
    if (tid == 0 && lse) lse[(int64_t)b * H + h] = M + __logf(L);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cls_attn_bwd_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                           const T* __restrict__ out, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, float* __restrict__ dq,
                                                           T* __restrict__ dkv, int Tk, int H) {
  __shared__ float sdq[SLOTS][64];
  const int tid = threadIdx.x, sub = tid & 7, slot = tid >> 3;
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const int D = H * 64;
  float qv[8], go[8], oo[8];
  Elem<T>::load8(q + (int64_t)b * D + h * 64 + sub * 8, qv);
  Elem<T>::load8(dout + (int64_t)b * D + h * 64 + sub * 8, go);
  Elem<T>::load8(out + (int64_t)b * D + h * 64 + sub * 8, oo);
  float delta = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) delta = fmaf(go[c], oo[c], delta);
  delta += dpp_move<0xB1>(delta);
  delta += dpp_move<0x4E>(delta);
  delta += dpp_move<0x141>(delta);
  const float L = lse[(int64_t)b * H + h];
  const T* kb = kv + (int64_t)b * Tk * 2 * D + h * 64 + sub * 8;
  T* db = dkv + (int64_t)b * Tk * 2 * D + h * 64 + sub * 8;
  float aq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = slot; j < Tk; j += SLOTS) {
    float kx[8], vx[8];
    Elem<T>::load8(kb + (int64_t)j * 2 * D, kx);
    Elem<T>::load8(kb + (int64_t)j * 2 * D + D, vx);
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      s = fmaf(qv[c], kx[c], s);
      dp = fmaf(go[c], vx[c], dp);
    }
    s += dpp_move<0xB1>(s);   dp += dpp_move<0xB1>(dp);
    s += dpp_move<0x4E>(s);   dp += dpp_move<0x4E>(dp);
    s += dpp_move<0x141>(s);  dp += dpp_move<0x141>(dp);
    const float p = __expf(0.125f * s - L);
    const float ds = p * (dp - delta) * 0.125f;        // d loss / d (q . k_j)
    float dk[8], dv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dk[c] = ds * qv[c];
      dv[c] = p * go[c];
      aq[c] = fmaf(ds, kx[c], aq[c]);
    }
    Elem<T>::store8(db + (int64_t)j * 2 * D, dk);
    Elem<T>::store8(db + (int64_t)j * 2 * D + D, dv);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) sdq[slot][sub * 8 + c] = aq[c];
  __syncthreads();
  if (tid < 64) {
    float a = 0.f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) a += sdq[s][tid];
    dq[(int64_t)b * D + h * 64 + tid] = a;
  }
}

}  // namespace

extern "C" int lvl_cls_attn_fwd(const void* q, const void* kv, void* out, float* lse, int B, int Tk, int H, int dtype,
                                void* stream) {
  LVL_REQUIRE(B == 0 || (q && kv && out && lse), "cls_attn_fwd: null pointer");
  LVL_REQUIRE(B >= 0 && Tk > 0 && H > 0, "cls_attn_fwd: bad shape B=%d T=%d H=%d", B, Tk, H);
  LVL_REQUIRE(lvl_aligned16(q) && lvl_aligned16(kv) && lvl_aligned16(out), "cls_attn_fwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cls_attn_fwd_kernel<T>), dim3((unsigned)(B * H)), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)q, (const T*)kv, (T*)out, lse, Tk, H, 1));
  LVL_CHECK_LAUNCH("cls_attn_fwd");
  return LVL_OK;
}

extern "C" int lvl_cross_attn_rows_fwd(const void* q, const void* kv, void* out, int rows, int qrep, int Tk, int H,
                                       int dtype, void* stream) {
  LVL_REQUIRE(rows == 0 || (q && kv && out), "cross_attn_rows_fwd: null pointer");
  LVL_REQUIRE(rows >= 0 && qrep > 0 && rows % qrep == 0 && Tk > 0 && H > 0,
              "cross_attn_rows_fwd: bad shape rows=%d qrep=%d T=%d H=%d", rows, qrep, Tk, H);
  LVL_REQUIRE((int64_t)rows * H < (1ll << 31), "cross_attn_rows_fwd: rows * heads must stay below 2^31");
  LVL_REQUIRE(lvl_aligned16(q) && lvl_aligned16(kv) && lvl_aligned16(out),
              "cross_attn_rows_fwd: pointers must be 16-byte aligned");
  if (rows == 0) return LVL_OK;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cls_attn_fwd_kernel<T>), dim3((unsigned)(rows * H)), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)q, (const T*)kv, (T*)out, (float*)nullptr,
                                               Tk, H, qrep));
  LVL_CHECK_LAUNCH("cross_attn_rows_fwd");
  return LVL_OK;
}

extern "C" int lvl_cls_attn_bwd(const void* q, const void* kv, const void* out, const void* dout, const float* lse,
                                float* dq, void* dkv, int B, int Tk, int H, int dtype, void* stream) {
  LVL_REQUIRE(B == 0 || (q && kv && out && dout && lse && dq && dkv), "cls_attn_bwd: null pointer");
  LVL_REQUIRE(B >= 0 && Tk > 0 && H > 0, "cls_attn_bwd: bad shape B=%d T=%d H=%d", B, Tk, H);
  LVL_REQUIRE(lvl_aligned16(q) && lvl_aligned16(kv) && lvl_aligned16(out) && lvl_aligned16(dout) && lvl_aligned16(dkv),
              "cls_attn_bwd: pointers must be 16-byte aligned");
  if (B == 0) return LVL_OK;
  LVL_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cls_attn_bwd_kernel<T>), dim3((unsigned)(B * H)), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)q, (const T*)kv, (const T*)out,
                                               (const T*)dout, lse, dq, (T*)dkv, Tk, H));
  LVL_CHECK_LAUNCH("cls_attn_bwd");
  return LVL_OK;
}
