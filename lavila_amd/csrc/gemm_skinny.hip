// y[M,N] = act(x[M,K] . w[N,K]^T + bias) for FEW rows (decoding: M = captions in flight), bf16 operands, gfx950.
//
// The persistent 256x256-tile GEMM (gemm_tn_mfma.hip) hands a whole 256-column weight panel to ONE compute unit: at
// M = 64 a GPT-2 Conv1D then costs 16-50 us however small it is (measured, profiles/r03_skinny_variants.json) -- one CU
// streams 0.4..1.5 MB of weights alone while 250 CUs idle. A decode step is ~100 such GEMMs. Here the work is cut into
// (16 rows) x (16 or 32 columns) outputs, one workgroup each, so that hundreds of compute units pull 50..200 KB each:
//   * the 8 waves of a workgroup split the CONTRACTION: wave w takes the 32-wide k-steps w, w+8, w+16, ... (adjacent
//     waves read adjacent 64-byte segments of every weight / activation row), keeps one f32 accumulator per column
//     block, and the eight partial sums meet in LDS at the end;
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHT rows as operand A and the activation rows as operand B: a lane then
//     holds 4 consecutive output columns of one row (8-byte stores), as in the big kernel;
//   * fragments go straight from memory to registers (16-byte loads, up to 6 k-steps in flight per wave before the first
//     MFMA). The first version of this kernel gave a workgroup all 64 rows of a strip: 0.5 MB through one CU's
//     texture path, 15 us per GEMM; 16 rows per workgroup cut that to the weights' share;
//   * wide matrices (N >= 2048) take 32 rows x 64 columns per workgroup: more reuse of every fragment while there are
//     still hundreds of workgroups (the choice per shape is measured, see the dispatch at the bottom); a wave's k-steps
//     come in PAIRS where that wins (both 64-byte halves of a 128-byte line);
//   * the 16-row blocks of one column strip re-read the strip's weights: the workgroup id is laid out so that they
//     run on the SAME XCD (id % 8) within 8 * row-blocks consecutive ids -- the re-reads hit that XCD's L2;
//   * epilogue: + bias, optional gelu_new / relu^2 (the two MLP activations of the gated GPT-2, gpt2_gated.py:363-396),
//     bf16 store. Rows >= M are never stored (their loads are clamped to row M-1).
// Beyond 128 rows, and for the lm_head's 50432 columns, the same entry point runs an LDS-staged tile kernel (mid_kernel
// below): there the strips' half-used cache lines and re-reads cost more than a barrier per K block.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 sk_bf16x8;
typedef __attribute__((ext_vector_type(4))) float sk_f32x4;

namespace {

constexpr int WAVES = 8;     // waves per workgroup = ways the contraction is split
// CH (template, default 6): k-steps fetched ahead per wave (6 steps x 8 waves = 1536 contraction elements per round)
constexpr int ACT_NONE = -1;

__device__ __forceinline__ sk_f32x4 sk_mfma(uint4 a, uint4 b, sk_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, a), __builtin_bit_cast(sk_bf16x8, b), c,
                                                 0, 0, 0);
}

template <int ACT>
__device__ __forceinline__ float sk_act(float v) {
  if (ACT == LVL_ACT_GELU_NEW) {
    // tanh(z) = 1 - 2 / (exp(2z) + 1) through v_exp_f32 / v_rcp_f32 (tanhf costs 2 us on a [640 x 3072] epilogue);
    // exp overflow -> 1, underflow -> -1, relative error ~1e-6 before the bf16 rounding
    const float z = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * z) + 1.f);
    return 0.5f * v * (1.f + t);
  }
  if (ACT == LVL_ACT_SQRELU) {
    const float t = fmaxf(v, 0.f);
    return t * t;
  }
  return v;
}

// The 8 partial accumulators of every output block meet in LDS; wave q (and q + 8, ...) finishes block (nb, rb) =
// (q / RB, q % RB): v[r] = y[row m0 + rb*16 + c][column n0 + nb*16 + g*4 + r] -> + bias, activation, bf16 store.
// F32O (round 5, the f32-class mode): x / w are bf16 TERM IMAGES of float32 operands ([M, 3K] = h|h|l, [N, 3K] = h|l|h, see
// lvl_split_bf16x3), the contraction over 3K accumulates the three products of the split, and the result leaves as
// float32 -- 16-byte stores, activation on the unrounded sum.
template <int NB, int RB, int ACT, bool F32O = false>
__device__ __forceinline__ void skinny_finish(sk_f32x4 (*part)[NB * RB][64], const sk_f32x4 (&acc)[NB][RB],
                                              const float* __restrict__ bias, uint16_t* __restrict__ y, int M, int N,
                                              int n0, int m0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) part[wave][nb * RB + rb][lane] = acc[nb][rb];
  __syncthreads();
#pragma unroll
  for (int q0 = 0; q0 < NB * RB; q0 += WAVES) {
    const int q = q0 + wave;
    const int nb = q / RB, rb = q % RB;
    const int row = m0 + rb * 16 + c;
    if (q < NB * RB && row < M) {
      sk_f32x4 v = part[0][q][lane];
#pragma unroll
      for (int ww = 1; ww < WAVES; ++ww) {
        const sk_f32x4 t = part[ww][q][lane];
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
      }
      const int n = n0 + nb * 16 + g * 4;
      if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (F32O) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (int64_t)row * N + n) =
            make_float4(sk_act<ACT>(v[0]), sk_act<ACT>(v[1]), sk_act<ACT>(v[2]), sk_act<ACT>(v[3]));
      } else {
        const uint2 o = make_uint2(f32x2_to_bf16x2(sk_act<ACT>(v[0]), sk_act<ACT>(v[1])),
                                   f32x2_to_bf16x2(sk_act<ACT>(v[2]), sk_act<ACT>(v[3])));
        *reinterpret_cast<uint2*>(y + (int64_t)row * N + n) = o;
      }
    }
  }
}

template <int NB, int RB, int ACT, bool PAIR, int CH = 6, bool F32O = false>
__global__ __launch_bounds__(512) void skinny_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                     const float* __restrict__ bias, uint16_t* __restrict__ y, int M,
                                                     int N, int K, int nstrips, int nmb) {
  extern __shared__ __align__(16) unsigned char sk_smem[];
  sk_f32x4 (*part)[NB * RB][64] = reinterpret_cast<sk_f32x4 (*)[NB * RB][64]>(sk_smem);   // [wave][block][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  // id = (strip / 8) * (8 * nmb) + mb * 8 + strip % 8: the row groups of a strip share an XCD and arrive together
  const int id = blockIdx.x;
  const int strip = (id / (8 * nmb)) * 8 + (id & 7), mb = (id >> 3) % nmb;
  if (strip >= nstrips) return;                            // uniform: the grid is padded to whole groups of 8 strips
  const int n0 = strip * (16 * NB), m0 = mb * (16 * RB);
  const uint16_t* wp[NB];
  const uint16_t* xp[RB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) wp[nb] = w + (int64_t)(n0 + nb * 16 + c) * K + g * 8;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int row = m0 + rb * 16 + c;
    xp[rb] = x + (int64_t)(row < M ? row : M - 1) * K + g * 8;
  }
  sk_f32x4 acc[NB][RB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[nb][rb] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = K >> 5;
  // step of slot i in a round starting at s0. PAIR: a wave takes both 64-byte halves of a 128-byte line (steps 2w, 2w+1)
  auto step_of = [&](int s0, int i) { return PAIR ? s0 + (i >> 1) * (2 * WAVES) + (i & 1) : s0 + WAVES * i; };
  for (int s0 = PAIR ? 2 * wave : wave; s0 < nsteps; s0 += WAVES * CH) {
    uint4 wf[CH][NB], xf[CH][RB];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int s = step_of(s0, i);
      if (s < nsteps) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wf[i][nb] = *reinterpret_cast<const uint4*>(wp[nb] + s * 32);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) xf[i][rb] = *reinterpret_cast<const uint4*>(xp[rb] + s * 32);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int s = step_of(s0, i);
      if (s < nsteps) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc[nb][rb] = sk_mfma(wf[i][nb], xf[i][rb], acc[nb][rb]);
      }
    }
  }
  skinny_finish<NB, RB, ACT, F32O>(part, acc, bias, y, M, N, n0, m0);
}

// ---- the same GEMM with the residual add and the LayerNorm of its INPUT in the prologue ---------------------------------
// x = LayerNorm(res + gate * y) * gamma + beta is what every LN-fed Conv1D of the decoder multiplies (q_attn, c_attn and
// the two c_fc of a block: gpt2_gated.py:441-487); as a kernel of its own that add + LayerNorm is 49 launches of ~4 us in
// a 1.2 ms decode step. A workgroup here owns whole rows (its waves split K, and K <= 1792 fits one round of fragments),
// so it forms the sum, the row statistics (two-pass, f32; partials across lanes by shuffles, across waves through LDS)
// and the normalised bf16 fragments in registers before its first MFMA. Every column strip repeats that for its rows
// (x is L2-resident and tiny); strip 0 also writes the new residual to res_out -- a DIFFERENT buffer than res, which the
// other strips are still reading.
struct LnPrologue {
  const uint16_t* y;        // [M, K] branch output to add, nullable
  const float* gate;        // one f32 (tanh(alpha)), nullable = 1
  const float* gamma;       // [K]
  const float* beta;        // [K]
  uint16_t* res_out;        // [M, K] = bf16(res + gate * y), nullable when y is null
  float eps;
};

__device__ __forceinline__ void sk_unpack(uint4 v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 sk_pack(const float (&f)[8]) {
  return make_uint4(f32x2_to_bf16x2(f[0], f[1]), f32x2_to_bf16x2(f[2], f[3]), f32x2_to_bf16x2(f[4], f[5]),
                    f32x2_to_bf16x2(f[6], f[7]));
}

template <int NB, int RB, int ACT, int SPW>      // SPW: k-steps per wave held in registers (K <= 256 * SPW)
__global__ __launch_bounds__(512) void skinny_ln_kernel(const uint16_t* __restrict__ res, const uint16_t* __restrict__ w,
                                                        const float* __restrict__ bias, uint16_t* __restrict__ out,
                                                        int M, int N, int K, int nstrips, int nmb, LnPrologue ln) {
  extern __shared__ __align__(16) unsigned char sk_smem[];
  sk_f32x4 (*part)[NB * RB][64] = reinterpret_cast<sk_f32x4 (*)[NB * RB][64]>(sk_smem);
  float (*red)[WAVES][RB][16] =                                           // [pass][wave][row block][row]
      reinterpret_cast<float (*)[WAVES][RB][16]>(sk_smem + sizeof(sk_f32x4) * WAVES * NB * RB * 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int id = blockIdx.x;
  const int strip = (id / (8 * nmb)) * 8 + (id & 7), mb = (id >> 3) % nmb;
  if (strip >= nstrips) return;
  const int n0 = strip * (16 * NB), m0 = mb * (16 * RB);
  const int nsteps = K >> 5;
  const float gt = ln.gate ? *ln.gate : 1.f;
  uint4 wf[SPW][NB], xf[SPW][RB];
  float psum[RB];
  // ---- loads: weights, residual rows, branch rows (steps wave, wave + 8, ...) ---------------------------------------------
  {
    uint4 yf[SPW][RB];
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      const int s = wave + WAVES * i;
      if (s < nsteps) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          wf[i][nb] = *reinterpret_cast<const uint4*>(w + (int64_t)(n0 + nb * 16 + c) * K + g * 8 + s * 32);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const int row = m0 + rb * 16 + c;
          const int64_t off = (int64_t)(row < M ? row : M - 1) * K + g * 8 + s * 32;
          xf[i][rb] = *reinterpret_cast<const uint4*>(res + off);
          if (ln.y) yf[i][rb] = *reinterpret_cast<const uint4*>(ln.y + off);
        }
      }
    }
    // ---- sum = bf16(res + gate * y): what the residual stream stores and what the LayerNorm sees ---------------------------
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) psum[rb] = 0.f;
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      const int s = wave + WAVES * i;
      if (s < nsteps) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          float f[8];
          sk_unpack(xf[i][rb], f);
          if (ln.y) {
            float t[8];
            sk_unpack(yf[i][rb], t);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = fmaf(gt, t[k], f[k]);
            xf[i][rb] = sk_pack(f);
            sk_unpack(xf[i][rb], f);                                       // the rounded values
            const int row = m0 + rb * 16 + c;
            if (strip == 0 && row < M)
              *reinterpret_cast<uint4*>(ln.res_out + (int64_t)row * K + g * 8 + s * 32) = xf[i][rb];
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) psum[rb] += f[k];
        }
      }
    }
  }
  // ---- row statistics: lanes c, c+16, c+32, c+48 hold the same row; then the 8 waves ---------------------------------------
  float mean[RB], rstd[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float v = psum[rb];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0) red[0][wave][rb][c] = v;
  }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww) t += red[0][ww][rb][c];
    mean[rb] = t / (float)K;
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) psum[rb] = 0.f;
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int s = wave + WAVES * i;
    if (s < nsteps) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        float f[8];
        sk_unpack(xf[i][rb], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float d = f[k] - mean[rb];
          psum[rb] = fmaf(d, d, psum[rb]);
        }
      }
    }
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float v = psum[rb];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0) red[1][wave][rb][c] = v;
  }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww) t += red[1][ww][rb][c];
    rstd[rb] = rsqrtf(t / (float)K + ln.eps);
  }
  // ---- normalise the fragments, multiply ------------------------------------------------------------------------------
  sk_f32x4 acc[NB][RB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[nb][rb] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    const int s = wave + WAVES * i;
    if (s < nsteps) {
      float ga[8], be[8];
      load8_f32(ln.gamma + s * 32 + g * 8, ga);
      load8_f32(ln.beta + s * 32 + g * 8, be);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        float f[8];
        sk_unpack(xf[i][rb], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = fmaf((f[k] - mean[rb]) * rstd[rb], ga[k], be[k]);
        const uint4 h = sk_pack(f);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb][rb] = sk_mfma(wf[i][nb], h, acc[nb][rb]);
      }
    }
  }
  skinny_finish<NB, RB, ACT>(part, acc, bias, out, M, N, n0, m0);
}

template <int NB, int RB, bool PAIR, int CHV = 6, bool F32O = false>
int launch_skinny(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, int act,
                  hipStream_t st) {
  const int nstrips = N / (16 * NB), nmb = (M + 16 * RB - 1) / (16 * RB);
  const dim3 grid((unsigned)(((nstrips + 7) / 8) * 8 * nmb));
  constexpr size_t lds = (size_t)WAVES * NB * RB * 64 * sizeof(sk_f32x4);
#define LVL_SK(A)                                                                                                 \
  do {                                                                                                            \
    if (lds > 64 * 1024)                                                                                          \
      if (int rc = lvl_allow_lds<skinny_kernel<NB, RB, A, PAIR, CHV, F32O>>()) return rc;                         \
    hipLaunchKernelGGL((skinny_kernel<NB, RB, A, PAIR, CHV, F32O>), grid, dim3(64 * WAVES), lds, st,              \
                       (const uint16_t*)x, (const uint16_t*)w, bias, (uint16_t*)y, M, N, K, nstrips, nmb);        \
  } while (0)
  if (act == LVL_ACT_GELU_NEW) LVL_SK(LVL_ACT_GELU_NEW);
  else if (act == LVL_ACT_SQRELU) LVL_SK(LVL_ACT_SQRELU);
  else LVL_SK(ACT_NONE);
#undef LVL_SK
  LVL_CHECK_LAUNCH("linear_skinny");
  return LVL_OK;
}

template <int NB, int RB, int SPW>
int launch_skinny_ln(const void* res, const void* w, const float* bias, void* out, int M, int N, int K, int act,
                     const LnPrologue& ln, hipStream_t st) {
  const int nstrips = N / (16 * NB), nmb = (M + 16 * RB - 1) / (16 * RB);
  const dim3 grid((unsigned)(((nstrips + 7) / 8) * 8 * nmb));
  constexpr size_t lds = (size_t)WAVES * NB * RB * 64 * sizeof(sk_f32x4) + 2 * WAVES * RB * 16 * sizeof(float);
#define LVL_SKL(A)                                                                                                \
  do {                                                                                                            \
    if (lds > 64 * 1024)                                                                                          \
      if (int rc = lvl_allow_lds<skinny_ln_kernel<NB, RB, A, SPW>>()) return rc;                                  \
    hipLaunchKernelGGL((skinny_ln_kernel<NB, RB, A, SPW>), grid, dim3(64 * WAVES), lds, st, (const uint16_t*)res, \
                       (const uint16_t*)w, bias, (uint16_t*)out, M, N, K, nstrips, nmb, ln);                      \
  } while (0)
  if (act == LVL_ACT_GELU_NEW) LVL_SKL(LVL_ACT_GELU_NEW);
  else if (act == LVL_ACT_SQRELU) LVL_SKL(LVL_ACT_SQRELU);
  else LVL_SKL(ACT_NONE);
#undef LVL_SKL
  LVL_CHECK_LAUNCH("linear_skinny_ln");
  return LVL_OK;
}

// ---- many rows (M > 128: 64 clips x 10 sampled captions, teacher-forced captions up to 8192 rows): operands through LDS ----
// The strip kernel's fragments come straight from memory: a wave instruction touches 16 rows x 64 bytes -- 16 half-used
// cache lines -- and a compute unit sustains only ~30 GB/s that way (measured: 17-19 us for the 640-row Conv1Ds where the
// library GEMM takes 8-12). With hundreds of rows the classic form pays: a workgroup tile of 16 RB WM x 16 NB WN outputs
// -- 64 x 128 (8 waves of 32 x 32) for wide matrices, 32 x 64 (4 waves of 16 x 32, two K groups) for narrow ones -- with K
// walked in blocks of 64:
//   * global -> registers with 8 lanes per 128-byte row (whole cache lines), DEPTH blocks in flight per thread (the
//     compiler's counted vmcnt waits keep DEPTH - 1 behind the one being written), then one ds_write_b128 per piece into
//     a double-buffered LDS tile whose 16-byte chunks are XOR-swizzled by the row;
//   * ds_read_b128 fragments in the MFMA layout (weights = operand A, rows of x = operand B, as everywhere in this file),
//     one barrier per K block; every wave owns its sub-tile for the whole K, so there is no cross-wave reduction;
//   * epilogue: + bias, gelu_new / relu^2, 8-byte bf16 stores.
template <int WM, int WN, int RB, int NB, int ACT, int DEPTH, int KG = 1>
__global__ __launch_bounds__(64 * WM * WN * KG) void mid_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                           const float* __restrict__ bias, uint16_t* __restrict__ y, int M,
                                                           int N, int K, int tiles_n, int tiles_m) {
  // KG = 2: two groups of WM x WN waves work on the SAME output tile, group q on the K blocks q, q + 2, ... with its own
  // LDS stages (a [768 x 3072] Conv1D at 640 rows is 120 tiles of 48 blocks: the chain, not the chip, is the limit);
  // their accumulators meet in LDS at the end.
  constexpr int NT = 64 * WM * WN;                              // threads of one K group
  constexpr int TM = 16 * RB * WM, TN = 16 * NB * WN;           // workgroup tile
  constexpr int XP = TM * 8 / NT, WP = TN * 8 / NT;             // 16-byte pieces per thread and K block
  static_assert(TM * 8 % NT == 0 && TN * 8 % NT == 0, "pieces divide evenly");
  extern __shared__ __align__(16) unsigned char sk_smem[];
  // stage s: x tile at s * (TM + TN) * 128, w tile behind it; rows of 128 bytes (64 contraction elements)
  const int lane = threadIdx.x & 63, wave_all = threadIdx.x >> 6;
  const int kg = wave_all / (WM * WN), wave = wave_all % (WM * WN), tid = threadIdx.x - kg * NT;
  const int c = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  // Workgroup i runs on XCD i % 8: each XCD takes a CONTIGUOUS range of the column-major tile order, so the row tiles
  // that share a weight panel sit behind one L2 and the weights cross the fabric about once, not once per XCD (PMC,
  // profiles/r03_narrator_traffic_n10.json: 22 MB per launch where 8-10 are algorithmic with the round-robin order).
  const int ntiles = tiles_n * tiles_m, per_xcd = (ntiles + 7) >> 3;
  const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (lin >= ntiles) return;                                   // uniform; the grid is 8 * per_xcd
  const int tn = lin / tiles_m, tm = lin - tn * tiles_m;
  const int m0 = tm * TM, n0 = tn * TN;
  // global sources of this thread's pieces: piece p -> tile row (p * NT + tid) / 8, 16-byte chunk tid % 8
  const uint16_t* xsrc[XP];
  const uint16_t* wsrc[WP];
  int xdst[XP], wdst[WP];
#pragma unroll
  for (int p = 0; p < XP; ++p) {
    const int r = (p * NT + tid) >> 3, ch = tid & 7;
    const int row = m0 + r;
    xsrc[p] = x + (int64_t)(row < M ? row : M - 1) * K + ch * 8;
    xdst[p] = r * 128 + ((ch ^ (r & 7)) << 4);
  }
#pragma unroll
  for (int p = 0; p < WP; ++p) {
    const int r = (p * NT + tid) >> 3, ch = tid & 7;
    const int col = n0 + r;
    wsrc[p] = w + (int64_t)(col < N ? col : N - 1) * K + ch * 8;
    wdst[p] = TM * 128 + r * 128 + ((ch ^ (r & 7)) << 4);
  }
  constexpr int STAGE = (TM + TN) * 128;
  sk_f32x4 acc[NB][RB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[nb][rb] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
  const int nblk_all = K >> 6;
  const int nblk = (nblk_all - kg + KG - 1) / KG;              // blocks of this K group: kg, kg + KG, ...
  const int nloop = (nblk_all + KG - 1) / KG;                  // iterations every wave runs (barriers are workgroup-wide)
  unsigned char* const stage_base = sk_smem + kg * 2 * STAGE;
  // DEPTH register sets of XP + WP pieces, filled and drained with compile-time indices only (native vector type: an
  // array of HIP's uint4 structs, or a set picked by a run-time slot number, ends up in scratch memory and serialises the
  // pipeline on its stores)
  lvl_u32x4 ring[DEPTH][XP + WP];
#define SK_FETCH(R, BLK)                                                                              \
  do {                                                                                                \
    const int blk__ = (BLK);                                                                          \
    _Pragma("unroll") for (int p = 0; p < XP; ++p)                                                    \
        R[p] = *reinterpret_cast<const lvl_u32x4*>(xsrc[p] + (blk__ * KG + kg) * 64);                 \
    _Pragma("unroll") for (int p = 0; p < WP; ++p)                                                    \
        R[XP + p] = *reinterpret_cast<const lvl_u32x4*>(wsrc[p] + (blk__ * KG + kg) * 64);            \
  } while (0)
#define SK_STASH(R, STG)                                                                              \
  do {                                                                                                \
    unsigned char* base__ = stage_base + (STG) * STAGE;                                               \
    _Pragma("unroll") for (int p = 0; p < XP; ++p) *reinterpret_cast<lvl_u32x4*>(base__ + xdst[p]) = R[p]; \
    _Pragma("unroll") for (int p = 0; p < WP; ++p)                                                    \
        *reinterpret_cast<lvl_u32x4*>(base__ + wdst[p]) = R[XP + p];                                  \
  } while (0)
  auto compute = [&](int stage) {
    const unsigned char* xs = stage_base + stage * STAGE;
    const unsigned char* ws = xs + TM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 wf[NB], xf[RB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int r = wn * (16 * NB) + nb * 16 + c;
        wf[nb] = *reinterpret_cast<const uint4*>(ws + r * 128 + (((ks * 4 + g) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int r = wm * (16 * RB) + rb * 16 + c;
        xf[rb] = *reinterpret_cast<const uint4*>(xs + r * 128 + (((ks * 4 + g) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[nb][rb] = sk_mfma(wf[nb], xf[rb], acc[nb][rb]);
    }
  };
  // prologue: DEPTH blocks requested, the first one staged. Every fetch and stash below is UNCONDITIONAL (block indices
  // are clamped to the last block, whose redundant copies nobody reads): the loads of a thread then retire in a fixed
  // order and the compiler's counted vmcnt waits let DEPTH - 1 blocks stay in flight behind the one being stashed.
  const int last = nblk > 0 ? nblk - 1 : 0;                    // (a group without blocks re-reads block 0 and never computes)
  auto clampb = [&](int b) { return b < last ? b : last; };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) SK_FETCH(ring[d], clampb(d));
  SK_STASH(ring[0], 0);
  __syncthreads();
  for (int k0 = 0; k0 < nloop; k0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {                      // d is a compile-time constant after unrolling
      const int k = k0 + d;
      SK_FETCH(ring[d], clampb(k + DEPTH));                // set d went to LDS one step ago
      if (k < nblk) compute(k & 1);
      SK_STASH(ring[(d + 1) % DEPTH], (k + 1) & 1);
      __syncthreads();
    }
  }
#undef SK_FETCH
#undef SK_STASH
  if (KG > 1) {                                               // the last barrier of the loop freed the stages
    sk_f32x4 (*part)[NB * RB][64] = reinterpret_cast<sk_f32x4 (*)[NB * RB][64]>(sk_smem);   // [(kg-1)*waves + wave][block][lane]
    if (kg > 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) part[(kg - 1) * (WM * WN) + wave][nb * RB + rb][lane] = acc[nb][rb];
    }
    __syncthreads();
    if (kg > 0) return;
#pragma unroll
    for (int q = 1; q < KG; ++q)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const sk_f32x4 t = part[(q - 1) * (WM * WN) + wave][nb * RB + rb][lane];
          acc[nb][rb][0] += t[0]; acc[nb][rb][1] += t[1]; acc[nb][rb][2] += t[2]; acc[nb][rb][3] += t[3];
        }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int row = m0 + wm * (16 * RB) + rb * 16 + c, n = n0 + wn * (16 * NB) + nb * 16 + g * 4;
      if (row < M && n < N) {
        sk_f32x4 v = acc[nb][rb];
        if (bias) {
          const float4 b = *reinterpret_cast<const float4*>(bias + n);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        const uint2 o = make_uint2(f32x2_to_bf16x2(sk_act<ACT>(v[0]), sk_act<ACT>(v[1])),
                                   f32x2_to_bf16x2(sk_act<ACT>(v[2]), sk_act<ACT>(v[3])));
        *reinterpret_cast<uint2*>(y + (int64_t)row * N + n) = o;
      }
    }
  }
}

template <int WM, int WN, int RB, int NB, int DEPTH, int KG = 1>
int launch_mid(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, int act, hipStream_t st) {
  constexpr int TM = 16 * RB * WM, TN = 16 * NB * WN;
  const int tiles_n = (N + TN - 1) / TN, tiles_m = (M + TM - 1) / TM;
  const dim3 grid((unsigned)(((tiles_n * tiles_m + 7) / 8) * 8));
  constexpr size_t stages = (size_t)KG * 2 * (TM + TN) * 128;
  constexpr size_t parts = (size_t)(KG - 1) * WM * WN * NB * RB * 64 * sizeof(sk_f32x4);
  constexpr size_t lds = stages > parts ? stages : parts;
#define LVL_MK(A)                                                                                                   \
  do {                                                                                                              \
    if (lds > 64 * 1024)                                                                                            \
      if (int rc = lvl_allow_lds<mid_kernel<WM, WN, RB, NB, A, DEPTH, KG>>()) return rc;                            \
    hipLaunchKernelGGL((mid_kernel<WM, WN, RB, NB, A, DEPTH, KG>), grid, dim3(64 * WM * WN * KG), lds, st,          \
                       (const uint16_t*)x, (const uint16_t*)w, bias, (uint16_t*)y, M, N, K, tiles_n, tiles_m);      \
  } while (0)
  if (act == LVL_ACT_GELU_NEW) LVL_MK(LVL_ACT_GELU_NEW);
  else if (act == LVL_ACT_SQRELU) LVL_MK(LVL_ACT_SQRELU);
  else LVL_MK(ACT_NONE);
#undef LVL_MK
  LVL_CHECK_LAUNCH("linear_skinny (mid)");
  return LVL_OK;
}

std::atomic<int> g_variant{0};      // lvl_debug_skinny_variant: 0 = the shipped choice

}  // namespace

extern "C" int lvl_linear_skinny(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, int act,
                                 void* stream) {
  LVL_REQUIRE(M == 0 || (x && w && y), "linear_skinny: null pointer");
  LVL_REQUIRE(M >= 0 && N > 0 && K > 0, "linear_skinny: bad shape M=%d N=%d K=%d", M, N, K);
  LVL_REQUIRE(act == -1 || act == LVL_ACT_GELU_NEW || act == LVL_ACT_SQRELU, "linear_skinny: unknown activation %d", act);
  if (N % 16 != 0 || K % 32 != 0 || (int64_t)(N / 16 + 8) * ((M + 15) / 16) >= (1ll << 31))
    return lvl_fail(LVL_ENOSYS, "linear_skinny: needs N %% 16 == 0 and K %% 32 == 0 (N=%d K=%d)", N, K);
  LVL_REQUIRE(lvl_aligned16(x) && lvl_aligned16(w) && lvl_aligned16(y) && lvl_aligned16(bias),
              "linear_skinny: pointers must be 16-byte aligned");
  if (M == 0) return LVL_OK;
  // Output block per workgroup, from measurements on the decoder's shapes (tools/probe_skinny.py,
  // profiles/r03_skinny_variants.json): up to 128 rows a 16 x 16 block with PAIRED k-steps (a wave reads whole 128-byte
  // lines: 12.6 -> 7.6 us at K = 3072), or 32 rows x 64 columns once N >= 2048 gives >= 64 strips; beyond 128 rows
  // (64 clips x 10 sampled captions = 640) 64 rows x 32 columns, which halves the re-reads of the weight strip
  // (25.6 -> 19.3 us for [768 x 3072]), or 64 x 64 for N >= 2048.
  const hipStream_t st = (hipStream_t)stream;
  switch (g_variant.load(std::memory_order_relaxed)) {       // measurement variants (tools/probe_skinny.py)
    case 1: if (N % 32 == 0) return launch_skinny<2, 2, true>(x, w, bias, y, M, N, K, act, st); break;
    case 2: if (N % 32 == 0) return launch_skinny<2, 4, false>(x, w, bias, y, M, N, K, act, st); break;
    case 3: if (N % 64 == 0) return launch_skinny<4, 4, false>(x, w, bias, y, M, N, K, act, st); break;
    case 4: if (N % 32 == 0) return launch_skinny<2, 4, true>(x, w, bias, y, M, N, K, act, st); break;
    case 5: if (N % 32 == 0) return launch_skinny<2, 1, true>(x, w, bias, y, M, N, K, act, st); break;
    case 6: return launch_skinny<1, 1, true>(x, w, bias, y, M, N, K, act, st);
    case 7: if (N % 64 == 0) return launch_skinny<4, 2, false>(x, w, bias, y, M, N, K, act, st); break;
    case 8: if (N % 32 == 0) return launch_skinny<2, 2, false>(x, w, bias, y, M, N, K, act, st); break;
    case 9: return launch_skinny<1, 1, false>(x, w, bias, y, M, N, K, act, st);
    case 10: if (N % 64 == 0) return launch_skinny<4, 4, false, 3>(x, w, bias, y, M, N, K, act, st); break;
    case 11: if (N % 64 == 0) return launch_skinny<4, 4, false, 4>(x, w, bias, y, M, N, K, act, st); break;
    case 12: if (N % 64 == 0) return launch_skinny<4, 4, true, 4>(x, w, bias, y, M, N, K, act, st); break;
    case 13: if (N % 32 == 0) return launch_skinny<2, 4, false, 4>(x, w, bias, y, M, N, K, act, st); break;
    case 14: if (K % 64 == 0) return launch_mid<2, 4, 2, 2, 5>(x, w, bias, y, M, N, K, act, st); break;   // 64 x 128, 5 blocks in flight
    case 15: if (K % 64 == 0) return launch_mid<2, 4, 2, 2, 3>(x, w, bias, y, M, N, K, act, st); break;   // 64 x 128 (8 waves)
    case 16: if (K % 64 == 0) return launch_mid<2, 2, 2, 2, 3>(x, w, bias, y, M, N, K, act, st); break;   // 64 x 64 (4 waves)
    case 17: if (K % 64 == 0) return launch_mid<2, 2, 2, 2, 6>(x, w, bias, y, M, N, K, act, st); break;   // 64 x 64, 6 blocks in flight
    case 18: if (K % 64 == 0) return launch_mid<2, 2, 2, 2, 3, 2>(x, w, bias, y, M, N, K, act, st); break;   // 64 x 64, two K groups
    case 19: if (K % 64 == 0) return launch_mid<2, 2, 1, 2, 3, 1>(x, w, bias, y, M, N, K, act, st); break;   // 32 x 64
    case 20: if (K % 128 == 0) return launch_mid<2, 2, 1, 2, 3, 2>(x, w, bias, y, M, N, K, act, st); break;  // 32 x 64, two K groups
    case 21: if (K % 128 == 0) return launch_mid<2, 2, 2, 1, 3, 2>(x, w, bias, y, M, N, K, act, st); break;  // 64 x 32, two K groups
    default: break;
  }
  if (K % 64 == 0) {
    // the LDS-staged kernel where whole-line loads and operand reuse decide (in-graph times, profiles/r03_skinny_variants.json):
    // lm_head [50432 x 768] at <= 128 rows 17.3 us (strips 40.6, the 256-column-panel kernel 31.6, library 16.3); beyond
    // 128 rows [3072 x 768] 8.6 us (strips 16.8, library 8.3), [2304 x 768] 8.3 (16.5, 8.0), [768 x 3072] 12.3 (19.3, 11.8)
    if (N >= 8192 && M <= 128) return launch_mid<2, 4, 2, 2, 3>(x, w, bias, y, M, N, K, act, st);       // 64 x 128 tiles
    if (M > 128) {
      if (N >= 2048) return launch_mid<2, 4, 2, 2, 3>(x, w, bias, y, M, N, K, act, st);                 // 64 x 128
      // narrow matrices: 32 x 64 tiles (240 workgroups for [640 x 768]), two K groups per tile: 5.3 us for [768 x 768]
      // (64 x 64 tiles 6.9, library 5.7), 12.3 us for [768 x 3072] (18.5 / 11.8)
      return K % 128 == 0 ? launch_mid<2, 2, 1, 2, 3, 2>(x, w, bias, y, M, N, K, act, st)
                          : launch_mid<2, 2, 1, 2, 3>(x, w, bias, y, M, N, K, act, st);
    }
  }
  if (M > 128) {
    // wide matrices: 64 x 64 outputs with 3 k-steps in flight (199 VGPRs) -- 16.6 vs 23.3 us for [3072 x 768] at 640 rows
    if (N >= 2048 && N % 64 == 0) return launch_skinny<4, 4, false, 3>(x, w, bias, y, M, N, K, act, st);
    return N % 32 == 0 ? launch_skinny<2, 4, false>(x, w, bias, y, M, N, K, act, st)
                       : launch_skinny<1, 2, false>(x, w, bias, y, M, N, K, act, st);
  }
  if (N >= 2048 && N % 64 == 0) return launch_skinny<4, 2, false>(x, w, bias, y, M, N, K, act, st);
  return launch_skinny<1, 1, true>(x, w, bias, y, M, N, K, act, st);
}

// f32-class mode (round 5): the narrator's float32 decoder and every float32 inference Linear whose widths the
// 256-column-panel kernel does not tile (reference: x @ W + b in float32, gpt2_gated.py:184-188,383-384). x3 [M, K3] and
// w3 [N, K3] are the bf16 term images of the float32 operands (K3 = 3 K, lvl_split_bf16x3 roles 0 / 1), y is float32.
// The strip kernels only (fragments straight from memory, float32 partial sums through LDS): this is the parity
// configuration, the shapes are small or the call is rare.
extern "C" int lvl_linear_skinny_f32c(const void* x3, const void* w3, const float* bias, float* y, int M, int N, int K3,
                                      int act, void* stream) {
  LVL_REQUIRE(M == 0 || (x3 && w3 && y), "linear_skinny_f32c: null pointer");
  LVL_REQUIRE(M >= 0 && N > 0 && K3 > 0, "linear_skinny_f32c: bad shape M=%d N=%d K3=%d", M, N, K3);
  LVL_REQUIRE(act == -1 || act == LVL_ACT_GELU_NEW || act == LVL_ACT_SQRELU, "linear_skinny_f32c: unknown activation %d", act);
  if (N % 16 != 0 || K3 % 96 != 0 || (int64_t)(N / 16 + 8) * ((M + 15) / 16) >= (1ll << 31))
    return lvl_fail(LVL_ENOSYS, "linear_skinny_f32c: needs N %% 16 == 0 and K3 = 3 K with K %% 32 == 0 (N=%d K3=%d)", N, K3);
  LVL_REQUIRE(lvl_aligned16(x3) && lvl_aligned16(w3) && lvl_aligned16(y) && lvl_aligned16(bias),
              "linear_skinny_f32c: pointers must be 16-byte aligned");
  if (M == 0) return LVL_OK;
  const hipStream_t st = (hipStream_t)stream;
  if (M > 128) {
    if (N % 64 == 0) return launch_skinny<4, 4, false, 3, true>(x3, w3, bias, y, M, N, K3, act, st);
    return N % 32 == 0 ? launch_skinny<2, 4, false, 6, true>(x3, w3, bias, y, M, N, K3, act, st)
                       : launch_skinny<1, 2, false, 6, true>(x3, w3, bias, y, M, N, K3, act, st);
  }
  if (N >= 2048 && N % 64 == 0) return launch_skinny<4, 2, false, 6, true>(x3, w3, bias, y, M, N, K3, act, st);
  return launch_skinny<1, 1, true, 6, true>(x3, w3, bias, y, M, N, K3, act, st);
}

extern "C" int lvl_debug_skinny_variant(int v) {
  g_variant.store(v, std::memory_order_relaxed);
  return LVL_OK;
}

extern "C" int lvl_linear_skinny_ln(const void* res, const void* y, const float* gate, const float* gamma,
                                    const float* beta, float eps, void* res_out, const void* w, const float* bias,
                                    void* out, int M, int N, int K, int act, void* stream) {
  LVL_REQUIRE(M == 0 || (res && gamma && beta && w && out), "linear_skinny_ln: null pointer");
  LVL_REQUIRE(!y || (res_out && res_out != res), "linear_skinny_ln: the new residual needs its own buffer (other strips still read res)");
  LVL_REQUIRE(M >= 0 && N > 0 && K > 0, "linear_skinny_ln: bad shape M=%d N=%d K=%d", M, N, K);
  LVL_REQUIRE(act == -1 || act == LVL_ACT_GELU_NEW || act == LVL_ACT_SQRELU, "linear_skinny_ln: unknown activation %d", act);
  if (N % 16 != 0 || K % 32 != 0 || K > 256 * 7 || (int64_t)(N / 16 + 8) * ((M + 15) / 16) >= (1ll << 31))
    return lvl_fail(LVL_ENOSYS, "linear_skinny_ln: needs N %% 16 == 0, K %% 32 == 0, K <= 1792 (N=%d K=%d)", N, K);
  LVL_REQUIRE(lvl_aligned16(res) && lvl_aligned16(y) && lvl_aligned16(res_out) && lvl_aligned16(w) && lvl_aligned16(out) &&
                  lvl_aligned16(bias) && lvl_aligned16(gamma) && lvl_aligned16(beta),
              "linear_skinny_ln: pointers must be 16-byte aligned");
  if (M == 0) return LVL_OK;
  const LnPrologue ln{(const uint16_t*)y, gate, gamma, beta, (uint16_t*)res_out, eps};
  const hipStream_t st = (hipStream_t)stream;
  const bool wide = N >= 2048 && N % 64 == 0;                 // same cut as lvl_linear_skinny up to 128 rows
  if (K <= 256 * 3)
    return wide ? launch_skinny_ln<4, 2, 3>(res, w, bias, out, M, N, K, act, ln, st)
                : launch_skinny_ln<1, 1, 3>(res, w, bias, out, M, N, K, act, ln, st);
  return launch_skinny_ln<1, 1, 7>(res, w, bias, out, M, N, K, act, ln, st);   // GPT-2 XL (K = 1600): 7 steps per wave
}
