// Forward and input-gradient GEMMs of the token-major Linear layers, with the element-wise neighbours of the
// reference fused into the epilogue:
//
//     Y[M,N] = epilogue( X[M,K] . W[N,K]^T ),   bf16 operands (both contraction-contiguous), f32 accumulation.
//
//   forward          y  = x W^T + b                    (W = the layer's weight [out,in])
//   input gradient   dx = dy Wt^T                      (Wt = the transposed bf16 weight copy [in,out] that
//                                                       lvl_cast_transpose writes beside the forward's cast)
// Epilogues (template EPI):
//   0  y = acc (+ bias)                                                    qkv / proj / fc2 / patch embed / dgrads
//   1  u = acc + bias -> aux_out ; y = u * sigmoid(1.702 u)                Mlp.fc1 + QuickGELU (timesformer.py:52-54)
//   2  y = acc * quickgelu'(aux_in) ; column sums of y -> partial slab     backward of the same: fc2's input gradient
//                                                                          becomes d(fc1 output); sums = d(fc1 bias)
//
// Shape of the problem on this path: M = B*T ~ 2e5 rows, N,K in {768, 2304, 3072}: the K = 768 products are close to
// the HBM ridge (384-614 flop/B), so the design is about streaming X once and never stalling the matrix pipe:
//   * 256x256 output tile per workgroup of 8 waves (2 along M x 4 along N), a wave owns 128x64 as 4x2 tiles of
//     v_mfma_f32_32x32x16_bf16 with SWAPPED operands (A = weight rows, B = activation rows): the accumulator then
//     holds, per lane, 4 consecutive output columns of one row, so bias / activation / aux tensors are read and
//     written as 8- and 16-byte vectors without an LDS transpose of the tile;
//   * K is walked 32 at a time through a 4-deep ring of LDS stages (X image 256 rows x 64 B | W image 256 rows x
//     64 B = 32 KiB per stage) filled by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) running 2-3 steps
//     ahead; one bare s_barrier per step, placed between the two K=16 halves of a step, so neither fill latency nor
//     LDS latency separates two steps' MFMAs (the schedule of wgrad_mfma.hip);
//   * the 16-byte chunks of a row are XOR-permuted inside the image (chunk ^ (-(row/4) & 3), applied to the per-lane
//     SOURCE address of the DMA and to the fragment reads) so that every ds_read_b128 lane group touches 16 distinct
//     16-byte slots of the 256-byte bank row: conflict-free for the 32-row fragments;
//   * blockIdx -> tile is XCD-aware (workgroup i runs on XCD i % 8): each XCD owns a contiguous range of the
//     N-fastest tile order, so the N/256 tiles that read the same 256 rows of X sit behind one L2 and X comes from
//     HBM once; W (<= 4.7 MB) lives in L2 / Infinity Cache.
#include <atomic>

#include "common.h"

// the LDS-DMA fills set M0 inside inline asm and say so in the clobber list; this kernel has no other M0 user
#pragma clang diagnostic ignored "-Winline-asm"

typedef __attribute__((ext_vector_type(8))) __bf16 gm_bf16x8;
typedef __attribute__((ext_vector_type(16))) float gm_f32x16;

namespace {

constexpr int BK = 32;                 // contraction elements per step (two 32x32x16 MFMAs deep)
constexpr int TM = 256, TN = 256;      // workgroup tile
constexpr int IMG = 256 * BK * 2;      // bytes of one operand image (256 rows x 64 B)
constexpr int STAGE_B = 2 * IMG;       // X image | W image
constexpr int NSTAGE = 4;
constexpr int NI = 4;                  // LDS-DMA fills per wave per step (32 fills of 1 KiB / 8 waves)

__device__ __forceinline__ gm_f32x16 mfma32(uint4 a, uint4 b, gm_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gm_bf16x8, a), __builtin_bit_cast(gm_bf16x8, b),
                                                 c, 0, 0, 0);
}

__device__ __forceinline__ float quick_gelu(float u) { return u / (1.f + __expf(-1.702f * u)); }
__device__ __forceinline__ float quick_gelu_grad(float u) {
  const float s = 1.f / (1.f + __expf(-1.702f * u));
  return s * (1.f + 1.702f * u * (1.f - s));
}

// lanes l < 32 and l + 32 hold adjacent 8-byte pieces (4 bf16) of the same output row for two neighbouring
// column groups `a` (columns c..c+3 | c+4..c+7) and `b` (c+8.. | c+12..): one half-swap per dword leaves the
// lower lane with 16 contiguous bytes of group a and the upper lane with 16 contiguous bytes of group b.
__device__ __forceinline__ uint4 widen_pair(uint2 a, uint2 b) {
  const auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
  return make_uint4(r0[0], r1[0], r0[1], r1[1]);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_tn_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                      const float* __restrict__ bias, uint16_t* __restrict__ Y,
                                                      uint16_t* __restrict__ aux_out,
                                                      const uint16_t* __restrict__ aux_in,
                                                      float* __restrict__ colpart, int64_t M, int N, int K,
                                                      int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];      // [NSTAGE][X image | W image]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // XCD-aware tile decode (bijective for any tile count): XCD x = bid % 8 owns a contiguous range of tiles
  const int bid = blockIdx.x, xcd = bid & 7;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int pair = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + (bid >> 3);
  const int tm = pair / tiles_n, tn = pair - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * TM;
  const int n0 = tn * TN;
  const int nsteps = K / BK;

  // ---- staging plan ---------------------------------------------------------------------------------------------
  // Fill f (0..31) of a step covers image rows 16*(f%16) .. +15 of X (f < 16) or W; lane l lands at stage byte
  // f*1024 + l*16 = row (l>>2), physical chunk (l&3), and therefore fetches the LOGICAL chunk (l&3) ^ g(row).
  const uint16_t* src[NI];
  uint32_t dst_off[NI];
  {
    const int r16 = lane >> 2;
    const int chunk = (lane & 3) ^ ((4 - (lane >> 4)) & 3);
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int f = wave + 8 * (q & 1);
      const int row = 16 * f + r16;
      if (q < 2) {
        int64_t gm = m0 + row;
        if (gm > M - 1) gm = M - 1;          // tail tile: re-read the last row, its products are never stored
        src[q] = X + gm * (int64_t)K + chunk * 8;
        dst_off[q] = (uint32_t)(f * 1024);
      } else {
        src[q] = W + (int64_t)(n0 + row) * K + chunk * 8;
        dst_off[q] = (uint32_t)(IMG + f * 1024);
      }
    }
  }
  // Fills go through inline asm: the compiler's LDS-DMA alias tracking would otherwise put s_waitcnt vmcnt(0) in
  // front of every LDS read and drain the run-ahead. Steps issued beyond the last one re-read the last K block.
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  int issued = 0;
  auto issue_loads = [&](int stage) {
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)stage * STAGE_B + dst_off[q]);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src[q])
                   : "memory", "m0");
      if (issued < nsteps - 1) src[q] += BK;
    }
    ++issued;
  };

  // ---- fragment addresses -----------------------------------------------------------------------------------------
  // 32x32x16 operand: lane l carries row (l & 31), contraction elements 8*(l>>5) .. +7 of the K=16 slice kk, i.e.
  // logical chunk 2*kk + (l>>5) of the row's four 16-byte chunks.
  const int r5 = lane & 31, hi = lane >> 5;
  const int g = (4 - ((r5 >> 2) & 3)) & 3;
  uint32_t offX[2], offW[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int p = (2 * kk + hi) ^ g;
    offX[kk] = (uint32_t)((wm * 128 + r5) * 64 + p * 16);
    offW[kk] = (uint32_t)(IMG + (wn * 64 + r5) * 64 + p * 16);
  }

  gm_f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto read_frags = [&](const uint8_t* st, int kk, uint4 (&xf)[4], uint4 (&wf)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const uint4*>(st + offW[kk] + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const uint4*>(st + offX[kk] + j * 2048);
  };
  auto multiply = [&](const uint4 (&xf)[4], const uint4 (&wf)[2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][j] = mfma32(wf[i], xf[j], acc[i][j]);
  };

  // Ring schedule, barrier in the MIDDLE of a step. Step s multiplies stage s%4 in two K=16 halves. The first
  // half's fragments were read during the previous step; while it runs, the second half's fragments are read.
  // Between the halves: wait until this wave's fills of step s+1 have landed (vmcnt counts them in order; the
  // fills of step s+2 stay in flight), s_barrier (=> step s+1 is complete for every wave and every wave is done
  // with stage s-1), issue the fills of step s+3 into the stage step s-1 used, read the first-half fragments of
  // step s+1, run the second half. The barrier is the bare s_barrier: a fence would drain the run-ahead.
  uint4 xp[4], wp[2], xq[4], wq[2];
  issue_loads(0);
  issue_loads(1);
  issue_loads(2);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(smem, 0, xp, wp);
  __builtin_amdgcn_sched_barrier(0);
  int stage = 0;
  for (int s = 0; s < nsteps; ++s) {
    const uint8_t* st = smem + stage * STAGE_B;
    const int nstage = (stage + 1) & (NSTAGE - 1);
    // sched_barrier(0): the machine scheduler would otherwise sink every fragment read down to its first use
    // (fewer live registers) and expose the LDS latency four times per step
    read_frags(st, 1, xq, wq);
    __builtin_amdgcn_sched_barrier(0);
    multiply(xp, wp);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_loads((stage + NSTAGE - 1) & (NSTAGE - 1));
    read_frags(smem + nstage * STAGE_B, 0, xp, wp);
    __builtin_amdgcn_sched_barrier(0);
    multiply(xq, wq);
    __builtin_amdgcn_sched_barrier(0);
    stage = nstage;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the run-ahead fills before this workgroup's LDS is freed

  // ---- epilogue -----------------------------------------------------------------------------------------------------
  // acc[i][j][4*rq + e] = output row m0 + wm*128 + j*32 + r5, column n0 + wn*64 + i*32 + 8*rq + 4*hi + e
  const int ncol0 = n0 + wn * 64 + 4 * hi;
  float csum[32];
  if (EPI == 2) {
#pragma unroll
    for (int c = 0; c < 32; ++c) csum[c] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t m = m0 + wm * 128 + j * 32 + r5;
    const bool valid = m < M;
    const int64_t mrow = valid ? m : M - 1;
    uint16_t* yrow = Y + mrow * (int64_t)N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint2 ypk[4], upk[4];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n = ncol0 + i * 32 + 8 * rq;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * rq + e];
        if (EPI != 2 && bias != nullptr) {
          const float4 b = *reinterpret_cast<const float4*>(bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (EPI == 1) {
          upk[rq] = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
          // the activation sees the ROUNDED pre-activation (what the reference's bf16 Linear output holds and what
          // the backward reads back)
          v[0] = quick_gelu(__uint_as_float(upk[rq].x << 16));
          v[1] = quick_gelu(__uint_as_float(upk[rq].x & 0xffff0000u));
          v[2] = quick_gelu(__uint_as_float(upk[rq].y << 16));
          v[3] = quick_gelu(__uint_as_float(upk[rq].y & 0xffff0000u));
        }
        if (EPI == 2) {
          const uint2 ub = *reinterpret_cast<const uint2*>(aux_in + mrow * (int64_t)N + n);
          v[0] *= quick_gelu_grad(__uint_as_float(ub.x << 16));
          v[1] *= quick_gelu_grad(__uint_as_float(ub.x & 0xffff0000u));
          v[2] *= quick_gelu_grad(__uint_as_float(ub.y << 16));
          v[3] *= quick_gelu_grad(__uint_as_float(ub.y & 0xffff0000u));
          if (valid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[i * 16 + rq * 4 + e] += v[e];
          }
        }
        ypk[rq] = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
      }
      // 16-byte stores: lower lanes take columns 16*jj .. +7, upper lanes 16*jj + 8 .. +15 of this 32-column group
      const int nst = n0 + wn * 64 + i * 32 + 8 * hi;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const uint4 yv = widen_pair(ypk[2 * jj], ypk[2 * jj + 1]);
        if (valid) *reinterpret_cast<uint4*>(yrow + nst + 16 * jj) = yv;
        if (EPI == 1) {
          const uint4 uv = widen_pair(upk[2 * jj], upk[2 * jj + 1]);
          if (valid) *reinterpret_cast<uint4*>(aux_out + mrow * (int64_t)N + nst + 16 * jj) = uv;
        }
      }
    }
  }
  if (EPI == 2) {
    // column sums over the wave's 128 rows: 32 values per lane, summed over the 32 lanes of each half-wave by a
    // halving butterfly (31 exchanges): lane r5 ends up with the total of value index c = r5
    int cnt = 16;
#pragma unroll
    for (int mask = 16; mask >= 1; mask >>= 1) {
      const bool up = (lane & mask) != 0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c < cnt) {
          const float lo = csum[c], hv = csum[c + cnt];
          const float send = up ? lo : hv, keep = up ? hv : lo;
          csum[c] = keep + __shfl_xor(send, mask, 64);
        }
      }
      cnt >>= 1;
    }
    const int c = r5;
    const int col = (c >> 4) * 32 + ((c >> 2) & 3) * 8 + 4 * hi + (c & 3);
    colpart[(size_t)(tm * 2 + wm) * N + n0 + wn * 64 + col] = csum[0];
  }
}

// out[n] = sum_p part[p][n]   (deterministic, no atomics): 64 columns x 16 row lanes per workgroup
__global__ __launch_bounds__(1024) void colpart_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int P, int N) {
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  float a = 0.f;
  if (n < N)
    for (int p = ry; p < P; p += 16) a += part[(size_t)p * N + n];
  red[ry][cx] = a;
  __syncthreads();
  if (ry == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cx];
    out[n] = t;
  }
}

// hipFuncSetAttribute once per (kernel instantiation, device), thread-safe
template <auto Kernel>
int allow_lds(int bytes) {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return lvl_fail(LVL_EHIP, "hipGetDevice failed");
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    const hipError_t e = hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return lvl_fail(LVL_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    done.fetch_or(bit, std::memory_order_release);
  }
  return LVL_OK;
}

template <int EPI>
int launch_tn(const void* x, const void* w, const float* bias, void* y, void* aux_out, const void* aux_in,
              float* colpart, int64_t M, int N, int K, hipStream_t st) {
  constexpr int shmem = NSTAGE * STAGE_B;
  const int rc = allow_lds<gemm_tn_kernel<EPI>>(shmem);
  if (rc != LVL_OK) return rc;
  const int tiles_n = N / TN;
  const int64_t tiles_m = (M + TM - 1) / TM;
  const int64_t ntiles = tiles_m * tiles_n;
  if (ntiles > 0x7fffffff) return lvl_fail(LVL_EINVAL, "linear_tn: too many tiles");
  hipLaunchKernelGGL((gemm_tn_kernel<EPI>), dim3((unsigned)ntiles), dim3(512), shmem, st, (const uint16_t*)x,
                     (const uint16_t*)w, bias, (uint16_t*)y, (uint16_t*)aux_out, (const uint16_t*)aux_in, colpart, M,
                     N, K, tiles_n, (int)ntiles);
  LVL_CHECK_LAUNCH("linear_tn");
  return LVL_OK;
}

}  // namespace

int64_t lvl_linear_tn_workspace_floats(int64_t M, int64_t N) { return 2 * ((M + TM - 1) / TM) * N; }

extern "C" int lvl_linear_tn(const void* x, const void* w, const float* bias, void* y, void* aux_out,
                             const void* aux_in, float* colsum, float* ws, int64_t M, int N, int K, int epilogue,
                             int dtype, void* stream) {
  LVL_REQUIRE(x && w && y, "linear_tn: null pointer");
  LVL_REQUIRE(dtype == LVL_BF16, "linear_tn: bf16 operands only (dtype=%d)", dtype);
  LVL_REQUIRE(M > 0 && N > 0 && K > 0, "linear_tn: empty problem");
  if (N % TN != 0 || K % BK != 0)
    return lvl_fail(LVL_ENOSYS, "linear_tn: no tiling for N=%d K=%d (N %% 256 == 0 and K %% 32 == 0 needed)", N, K);
  LVL_REQUIRE(lvl_aligned16(x) && lvl_aligned16(w) && lvl_aligned16(y) && lvl_aligned16(bias) &&
                  lvl_aligned16(aux_out) && lvl_aligned16(aux_in),
              "linear_tn: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case LVL_EPI_BIAS:
      return launch_tn<0>(x, w, bias, y, nullptr, nullptr, nullptr, M, N, K, st);
    case LVL_EPI_BIAS_QUICKGELU:
      LVL_REQUIRE(aux_out != nullptr, "linear_tn: the QuickGELU epilogue writes the pre-activation to aux_out");
      return launch_tn<1>(x, w, bias, y, aux_out, nullptr, nullptr, M, N, K, st);
    case LVL_EPI_QUICKGELU_BWD: {
      LVL_REQUIRE(aux_in != nullptr && colsum != nullptr && ws != nullptr && lvl_aligned16(ws),
                  "linear_tn: the QuickGELU-backward epilogue needs aux_in, colsum and a workspace");
      const int rc = launch_tn<2>(x, w, nullptr, y, nullptr, aux_in, ws, M, N, K, st);
      if (rc != LVL_OK) return rc;
      const int P = (int)(2 * ((M + TM - 1) / TM));
      hipLaunchKernelGGL(colpart_reduce_kernel, dim3((unsigned)((N + 63) / 64)), dim3(1024), 0, st, ws, colsum, P, N);
      LVL_CHECK_LAUNCH("linear_tn_colsum");
      return LVL_OK;
    }
    default:
      return lvl_fail(LVL_EINVAL, "linear_tn: unknown epilogue %d", epilogue);
  }
}
