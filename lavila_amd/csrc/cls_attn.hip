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

// qrep >= 2 query rows per context (the sampled captions of one clip, or the positions of a teacher-forced caption): one
// workgroup per (context, head) stages that head's keys and values in LDS ONCE (Tk x 64 x 2 elements: 64 KB for the
// narrator's 256 image tokens in bf16) and each of its NW waves then serves query rows on its own -- 8 key slots x 8 lanes,
// flash-style running (max, sum, acc) per slot, the slots merged by lane exchanges, no block barrier per row.
// Measured at 64 clips x 12 heads x 256 tokens (50 MB of keys / values per call, from HBM): 12 us for one row per clip
// on the direct kernel (4.1 TB/s), 19 / 32 / 46 us for 2 / 10 / 20 rows per clip here vs 24 / 40 / 80 re-reading per row.
// The per-row part is VALU work (dot products on 8 lanes per key); an MFMA formulation (rows padded to 16) is the next step.
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void cross_attn_shared_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                                    T* __restrict__ out, int Tk, int H, int qrep) {
  extern __shared__ __align__(16) unsigned char ca_smem[];
  T* ks = reinterpret_cast<T*>(ca_smem);
  T* vs = ks + (size_t)Tk * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 7, slot = lane >> 3;
  const int h = blockIdx.x % H, ctx = blockIdx.x / H;
  const int D = H * 64;
  const T* kb = kv + (int64_t)ctx * Tk * 2 * D + h * 64;
  // stage K and V: 8 loads in flight per thread before the first LDS store; 16-byte slots XOR-swizzled by the row so that
  // the 8 key slots of a wave (consecutive rows, same channel slot) hit different banks
  constexpr int NT = 64 * NW;
  for (int base = 0; base < Tk * 8; base += NT * 4) {
    RawVec<T, 8> kr[4], vr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * NT + tid;
      if (idx < Tk * 8) {
        const int j = idx >> 3, seg = (idx & 7) * 8;
        kr[u].load(kb + (int64_t)j * 2 * D + seg);
        vr[u].load(kb + (int64_t)j * 2 * D + D + seg);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * NT + tid;
      if (idx < Tk * 8) {
        const int j = idx >> 3, off = j * 64 + (((idx & 7) ^ (j & 7)) << 3);
        float t[8];
        kr[u].unpack(t);
        Elem<T>::store8(ks + off, t);
        vr[u].unpack(t);
        Elem<T>::store8(vs + off, t);
      }
    }
  }
  __syncthreads();
  for (int r = wave; r < qrep; r += NW) {
    const int64_t row = (int64_t)ctx * qrep + r;
    float qv[8];
    Elem<T>::load8(q + row * D + h * 64 + sub * 8, qv);
#pragma unroll
    for (int c = 0; c < 8; ++c) qv[c] *= 0.125f;
    float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // 4 keys per slot and step: four independent score chains (LDS read -> dot -> lane reduce) hide each other's
    // latency -- one wave per SIMD has nothing else to switch to -- and share one running-max update
    for (int j0 = slot; j0 < Tk; j0 += 32) {
      float sc4[4], vx[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + 8 * u;
        sc4[u] = -INFINITY;
        if (j < Tk) {
          float kx[8];
          const int off = j * 64 + ((sub ^ (j & 7)) << 3);
          Elem<T>::load8(ks + off, kx);
          Elem<T>::load8(vs + off, vx[u]);
          float s = qv[0] * kx[0];
#pragma unroll
          for (int c = 1; c < 8; ++c) s = fmaf(qv[c], kx[c], s);
          s += dpp_move<0xB1>(s);
          s += dpp_move<0x4E>(s);
          s += dpp_move<0x141>(s);
          sc4[u] = s;
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) vx[u][c] = 0.f;
        }
      }
      const float mn = fmaxf(fmaxf(m, sc4[0]), fmaxf(fmaxf(sc4[1], sc4[2]), sc4[3]));   // sc4[0] is always a real key
      const float corr = __expf(m - mn);
      float p4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p4[u] = __expf(sc4[u] - mn);
      l = l * corr + ((p4[0] + p4[1]) + (p4[2] + p4[3]));
#pragma unroll
      for (int c = 0; c < 8; ++c)
        acc[c] = fmaf(p4[0], vx[0][c], fmaf(p4[1], vx[1][c], fmaf(p4[2], vx[2][c], fmaf(p4[3], vx[3][c], acc[c] * corr))));
      m = mn;
    }
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {                     // merge the 8 slots (lanes differing in bits 3..5)
      const float m2 = __shfl_xor(m, d, 64), l2 = __shfl_xor(l, d, 64);
      const float mn = fmaxf(m, m2);
      const float a = m == -INFINITY ? 0.f : __expf(m - mn), b = m2 == -INFINITY ? 0.f : __expf(m2 - mn);
      l = l * a + l2 * b;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = acc[c] * a + __shfl_xor(acc[c], d, 64) * b;
      m = mn;
    }
    if (slot == 0) {
      const float inv = 1.f / l;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] *= inv;
      Elem<T>::store8(out + row * D + h * 64 + sub * 8, acc);
    }
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

namespace { std::atomic<int> g_shared_waves{0}; }

int lvl_launch_cross_attn_mfma(const void* q, const void* kv, void* out, int contexts, int qrep, int Tk, int H,
                               hipStream_t st);          // cross_attn_mfma.hip

// measurement hook (tools/probe_decode_kernels.py): waves per workgroup of the shared-context kernel (0 = by qrep)
extern "C" int lvl_debug_cross_attn_waves(int waves) {
  g_shared_waves.store(waves, std::memory_order_relaxed);
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
  // bf16, several rows per context, <= 256 keys: the MFMA kernel (lvl_debug_cross_attn_waves(n != 0) keeps the VALU form)
  if (dtype == LVL_BF16 && qrep >= 2 && Tk <= 256 && g_shared_waves.load(std::memory_order_relaxed) == 0)
    return lvl_launch_cross_attn_mfma(q, kv, out, rows / qrep, qrep, Tk, H, (hipStream_t)stream);
  const size_t lds = (size_t)Tk * 128 * (dtype == LVL_F32 ? 4 : 2);
  if (qrep >= 2 && lds <= 150 * 1024) {           // the context's keys / values fit LDS: read them once per (context, head)
    const unsigned grid = (unsigned)(rows / qrep * H);
    int nw = g_shared_waves.load(std::memory_order_relaxed);
    if (nw <= 0) nw = 16;      // measured (profiles/r03_decode_kernels.json): 16 waves win from qrep = 2 on (19 vs 25 us)
#define LVL_CA(TT, NWV)                                                                                       \
  do {                                                                                                        \
    if (lds > 64 * 1024)                                                                                      \
      if (int rc = lvl_allow_lds<cross_attn_shared_kernel<TT, NWV>>()) return rc;                             \
    hipLaunchKernelGGL((cross_attn_shared_kernel<TT, NWV>), dim3(grid), dim3(64 * NWV), lds, (hipStream_t)stream, \
                       (const TT*)q, (const TT*)kv, (TT*)out, Tk, H, qrep);                                   \
  } while (0)
    if (dtype == LVL_F32) {
      if (nw >= 16) LVL_CA(float, 16); else if (nw >= 8) LVL_CA(float, 8); else LVL_CA(float, 4);
    } else if (dtype == LVL_BF16) {
      if (nw >= 16) LVL_CA(bf16_t, 16); else if (nw >= 8) LVL_CA(bf16_t, 8); else LVL_CA(bf16_t, 4);
    } else {
      return lvl_fail(LVL_EINVAL, "unknown dtype %d", dtype);
    }
#undef LVL_CA
    LVL_CHECK_LAUNCH("cross_attn_rows_fwd");
    return LVL_OK;
  }
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
