// Decoder cross-attention for SEVERAL query rows per context on MFMA, bf16, gfx950 (lvl_cross_attn_rows_fwd, qrep >= 2:
// the sampled captions of one clip while decoding, or the positions of a teacher-forced caption;
// gpt2_gated.py:206-238,327-334 without a mask). One workgroup per (context, head): the head's keys and values
// (<= 256 x 64 each) are staged ONCE into the swizzled LDS images of the training kernels (attn_mfma_common.h), then
// each wave takes 16 query rows at a time:
//   S^T = K . Q^T      v_mfma_f32_16x16x32_bf16, K fragments from LDS, Q fragments straight from memory; all 16 key tiles
//                      stay in registers (64 VGPRs), so the softmax is exact single-pass
//   O   = P . V        P packed to bf16 in registers (two key tiles = one 32-deep contraction), V through the LDS
//                      transpose read (ds_read_tr16_b64): no transposed copy is staged
// The VALU form (cls_attn.hip) spends 1.3 us per extra query row on 8-lane dot products; here a row tile costs 64 MFMAs.
#include "attn_mfma_common.h"

using namespace attn_mfma;

namespace {

constexpr int XW = 4;          // waves per workgroup
constexpr int NKT = 16;        // key tiles of 16: up to 256 keys (the narrator pools every clip onto 256 image tokens)

__global__ __launch_bounds__(64 * XW) void cross_attn_mfma_kernel(const uint16_t* __restrict__ q,
                                                                  const uint16_t* __restrict__ kv,
                                                                  uint16_t* __restrict__ out, int Tk, int H, int qrep) {
  extern __shared__ __align__(16) uint16_t xa_smem[];
  uint16_t* Ks = xa_smem;
  uint16_t* Vs = Ks + NKT * 16 * RS;
  uint16_t* Ot = Vs + NKT * 16 * RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % H, ctx = blockIdx.x / H;
  const int D = H * 64;
  const uint16_t* kb = kv + (size_t)ctx * Tk * 2 * D + h * 64;
  stage_rows2<PrecBf16, 64 * XW, NKT / 2>(Ks, kb, (size_t)2 * D, nullptr, Vs, kb + D, (size_t)2 * D, nullptr, NKT * 16, Tk, tid, 0);
  __syncthreads();
  constexpr float kScale = 0.125f, kExp2 = 0.125f * 1.4426950408889634f;
  const FragOff fo = frag_offsets(lane);
  uint16_t* ot = Ot + wave * 16 * OS;
  const int ntiles = (qrep + 15) >> 4;
#pragma unroll 1
  for (int qt = wave; qt < ntiles; qt += XW) {
    const int qrow = qt * 16 + c;
    const uint16_t* qp = q + ((size_t)ctx * qrep + (qrow < qrep ? qrow : qrep - 1)) * D + h * 64 + g * 8;
    const uint4 qf0 = *reinterpret_cast<const uint4*>(qp);
    const uint4 qf1 = *reinterpret_cast<const uint4*>(qp + 32);
    f32x4 acc[NKT];
#pragma unroll
    for (int k = 0; k < NKT; ++k) acc[k] = mfma(tile_frag(Ks, k, fo.a[0]), qf0, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int k = 0; k < NKT; ++k) acc[k] = mfma(tile_frag(Ks, k, fo.a[1]), qf1, acc[k]);
    // acc[k][r] = raw S[query c][key k*16 + g*4 + r]
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NKT; ++k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k * 16 + g * 4 + r;
        acc[k][r] = key < Tk ? acc[k][r] : -INFINITY;
        m = fmaxf(m, acc[k][r]);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mk = m * kExp2;                    // key 0 always exists: m is finite
    float l = 0.f;
#pragma unroll
    for (int k = 0; k < NKT; ++k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(acc[k][r], kExp2, -mk));
        acc[k][r] = p;
        l += p;
      }
    }
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NKT / 2; ++j) {
      uint4 pa;
      pa.x = pack_bf16x2(acc[2 * j][0], acc[2 * j][1]);
      pa.y = pack_bf16x2(acc[2 * j][2], acc[2 * j][3]);
      pa.z = pack_bf16x2(acc[2 * j + 1][0], acc[2 * j + 1][1]);
      pa.w = pack_bf16x2(acc[2 * j + 1][2], acc[2 * j + 1][3]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint2 lo = tile_frag_tr(Vs, 2 * j, fo.tr[dt]);
        const uint2 hi = tile_frag_tr(Vs, 2 * j + 1, fo.tr[dt]);
        o[dt] = mfma(pa, make_uint4(lo.x, lo.y, hi.x, hi.y), o[dt]);
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    // o[dt][r] = O[query g*4+r][channel dt*16 + c]: normalise, transpose through the wave's LDS tile, store whole rows
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float linv = __builtin_amdgcn_rcpf(__shfl(l, g * 4 + r, 64));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ot[(g * 4 + r) * OS + dt * 16 + c] = f32_to_bf16(o[dt][r] * linv);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {                  // same wave wrote and reads: in order, no barrier
      const int row = (lane >> 3) + 8 * k, ch = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(ot + row * OS + ch * 8);
      const int qq = qt * 16 + row;
      if (qq < qrep) *reinterpret_cast<uint4*>(out + ((size_t)ctx * qrep + qq) * D + h * 64 + ch * 8) = v;
    }
    (void)kScale;
  }
}

}  // namespace

// called by lvl_cross_attn_rows_fwd (cls_attn.hip) for bf16, qrep >= 2, Tk <= 256
int lvl_launch_cross_attn_mfma(const void* q, const void* kv, void* out, int contexts, int qrep, int Tk, int H,
                               hipStream_t st) {
  const size_t lds = ((size_t)2 * NKT * 16 * RS + (size_t)XW * 16 * OS) * sizeof(uint16_t);
  if (int rc = lvl_allow_lds<cross_attn_mfma_kernel>()) return rc;
  hipLaunchKernelGGL(cross_attn_mfma_kernel, dim3((unsigned)(contexts * H)), dim3(64 * XW), lds, st, (const uint16_t*)q,
                     (const uint16_t*)kv, (uint16_t*)out, Tk, H, qrep);
  LVL_CHECK_LAUNCH("cross_attn_rows_fwd (mfma)");
  return LVL_OK;
}
