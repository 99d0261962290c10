// Space-mode divided attention forward on the matrix cores (bf16 in, f32 accumulate), gfx950.
//
// One 256-thread workgroup per (sample b, frame f, head h): N patch queries x (1 cls + N patch) keys,
// head dim 64 (timesformer.py:116-131 with the '(b f) n d' grouping :300-301). All keys of the group
// are LDS-resident, so the softmax is exact single-pass (no online rescale):
//   stage   K rows -> Ks[key][80]      (row-major, 160-B stride: conflict-free ds_read_b128 A-fragments)
//           V rows -> Vt[d][keys+8]    (transposed, written as packed row pairs; B-fragments of P.V are
//                                       two conflict-free ds_read_b64 of 4 consecutive keys)
//   S^T = K . Q^T per 16-query tile (v_mfma_f32_16x16x32_bf16; Q fragments straight from HBM: each
//           query row is used by exactly one wave). In the C layout every lane owns ONE query column,
//           so max/sum are in-lane plus two xor-shuffles, and the exponentiated tile is already the
//           A operand of O = P.V (k-order permuted identically on the V side): no cross-lane traffic.
//   O tile -> per-wave LDS transpose -> 16-B row-contiguous stores.
// The CLS query (token 0) attends to ALL keys; each workgroup adds the flash-style partial
// (max, sum, acc[64]) over its own frame's keys from the same LDS image (the cls key itself is taken
// by frame 0), and a tiny combine kernel merges the F partials: K and V are read from HBM once.
//
// Roofline: algorithmic HBM bytes per (b,f,h) = 4 * N * 64 * 2 (q,k,v in, o out); MFMA work is ~1/3 of
// the HBM time at 8 TB/s on TSF-B, so the kernel is built to stream: 2
// workgroups per CU (<= 80 KB LDS each) overlap one group's staging with the other's MFMA phase.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int KS = 80;            // Ks row stride in elements (160 B)
constexpr int OS = 72;            // per-wave output tile row stride in elements (144 B)
constexpr int CLS_REC = 66;       // cls partial record: m, l, acc[64]

template <int NKT> struct SpaceLds {
  static constexpr int KROWS = NKT * 16;
  static constexpr int LDK = NKT * 16 + 8;              // LDK/2 = 4*odd: conflict-free b64 reads
  static constexpr int ks_off = 0;                                       // bytes
  static constexpr int vt_off = ks_off + KROWS * KS * 2;
  static constexpr int ot_off = vt_off + 64 * LDK * 2;                   // 4 waves x [16][OS]
  static constexpr int qc_off = ot_off + 4 * 16 * OS * 2;               // f32[64]
  static constexpr int sc_off = qc_off + 64 * 4;                         // f32[KROWS]
  static constexpr int red_off = sc_off + KROWS * 4;                     // f32[4][64] + f32[8]
  static constexpr int total = red_off + (4 * 64 + 8) * 4;
};

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// TEXT = true reuses the kernel for the causal text tower (openai_model.py:196-198): one group per (b, h),
// L queries x L keys, no cls row, key j visible to query i iff j <= i, no CLS partial.
template <int NKT, bool TEXT>
__global__ __launch_bounds__(256, 2) void space_fwd_kernel(const uint16_t* __restrict__ qkv,
                                                           uint16_t* __restrict__ out, float* __restrict__ lse,
                                                           float* __restrict__ cls_ws, int F, int N, int H) {
  using L = SpaceLds<NKT>;
  constexpr int LDK = L::LDK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* Ks = reinterpret_cast<uint16_t*>(smem + L::ks_off);
  uint16_t* Vt = reinterpret_cast<uint16_t*>(smem + L::vt_off);
  uint16_t* Ot = reinterpret_cast<uint16_t*>(smem + L::ot_off);
  float* qc = reinterpret_cast<float*>(smem + L::qc_off);
  float* sc = reinterpret_cast<float*>(smem + L::sc_off);
  float* red = reinterpret_cast<float*>(smem + L::red_off);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x % H, f = (blockIdx.x / H) % F, b = blockIdx.x / (H * F);
  const int D = H * 64, T = TEXT ? N : 1 + F * N, nkeys = TEXT ? N : N + 1;
  const size_t tstride = (size_t)3 * D;
  const uint16_t* base = qkv + (size_t)b * T * tstride + h * 64;      // + token*3D (+D: k, +2D: v)
  const int tok0 = TEXT ? 0 : 1 + f * N;                               // token of query 0 (and of key row 1)

  // ---- stage K (row-major) and V (transposed, packed row pairs) ------------------------------------
  {
    const int c8 = tid & 7, r_in = tid >> 3, par = r_in & 1, rot = c8 & 3;
#pragma unroll 1
    for (int r0 = 0; r0 < L::KROWS; r0 += 32) {
      const int r = r0 + r_in;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (r < nkeys) {
        const uint16_t* p = base + (size_t)(TEXT ? r : (r == 0 ? 0 : tok0 + r - 1)) * tstride + c8 * 8;
        kv = *reinterpret_cast<const uint4*>(p + D);
        vv = *reinterpret_cast<const uint4*>(p + 2 * D);
      }
      if (r < L::KROWS) *reinterpret_cast<uint4*>(Ks + r * KS + c8 * 8) = kv;
      // even row writes channels c8*8+0..3 of the pair, odd row channels +4..7: swap the halves needed
      const uint32_t s0 = par ? vv.x : vv.z, s1 = par ? vv.y : vv.w;
      const uint32_t p0 = __shfl_xor(s0, 8, 64), p1 = __shfl_xor(s1, 8, 64);
      const uint32_t o0 = par ? vv.z : vv.x, o1 = par ? vv.w : vv.y;
      const uint32_t lo0 = par ? p0 : o0, lo1 = par ? p1 : o1;          // even row's two dwords
      const uint32_t hi0 = par ? o0 : p0, hi1 = par ? o1 : p1;          // odd row's two dwords
      uint32_t pk0 = (lo0 & 0xffffu) | (hi0 << 16), pk1 = (lo0 >> 16) | (hi0 & 0xffff0000u);
      uint32_t pk2 = (lo1 & 0xffffu) | (hi1 << 16), pk3 = (lo1 >> 16) | (hi1 & 0xffff0000u);
      // rotate by rot so that the 8 c8-lanes of a row hit different banks at every step (2-way at most)
      uint32_t t0 = (rot & 1) ? pk1 : pk0, t1 = (rot & 1) ? pk2 : pk1, t2 = (rot & 1) ? pk3 : pk2,
               t3 = (rot & 1) ? pk0 : pk3;
      uint32_t w0 = (rot & 2) ? t2 : t0, w1 = (rot & 2) ? t3 : t1, w2 = (rot & 2) ? t0 : t2,
               w3 = (rot & 2) ? t1 : t3;
      if (r < L::KROWS) {
        uint16_t* col = Vt + (size_t)(c8 * 8 + 4 * par) * LDK + (r & ~1);
        *reinterpret_cast<uint32_t*>(col + ((0 + rot) & 3) * LDK) = w0;
        *reinterpret_cast<uint32_t*>(col + ((1 + rot) & 3) * LDK) = w1;
        *reinterpret_cast<uint32_t*>(col + ((2 + rot) & 3) * LDK) = w2;
        *reinterpret_cast<uint32_t*>(col + ((3 + rot) & 3) * LDK) = w3;
      }
    }
    if (tid < 64) qc[tid] = bf16_to_f32(base[tid]) * 0.125f;           // cls query of this head, pre-scaled
  }
  __syncthreads();

  // ---- patch queries: one 16-query tile per wave at a time ------------------------------------------------
  const int c = lane & 15, g = lane >> 4;
  uint16_t* ot = Ot + wave * 16 * OS;
#pragma unroll 1
  for (int qt = wave; qt * 16 < N; qt += 4) {
    const int qrow = qt * 16 + c;
    const uint16_t* qp = base + (size_t)(tok0 + (qrow < N ? qrow : N - 1)) * tstride + g * 8;
    const bf16x8 qf0 = as_bf16x8(*reinterpret_cast<const uint4*>(qp));
    const bf16x8 qf1 = as_bf16x8(*reinterpret_cast<const uint4*>(qp + 32));
    f32x4 acc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const uint16_t* kp = Ks + (kt * 16 + c) * KS + g * 8;
      const bf16x8 a0 = as_bf16x8(*reinterpret_cast<const uint4*>(kp));
      const bf16x8 a1 = as_bf16x8(*reinterpret_cast<const uint4*>(kp + 32));
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf0, z, 0, 0, 0);
      acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf1, z, 0, 0, 0);
    }
    // acc[kt][r] = S[query c][key kt*16 + g*4 + r]
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + g * 4 + r;
        const bool vis = key < nkeys && (!TEXT || key <= qrow);
        const float s = vis ? acc[kt][r] * 0.125f : -INFINITY;
        acc[kt][r] = s;
        m = fmaxf(m, s);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(acc[kt][r] - m);
        acc[kt][r] = p;
        l += p;
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (NKT + 1) / 2; ++j) {
      uint4 pa;
      pa.x = pack_bf16x2(acc[2 * j][0], acc[2 * j][1]);
      pa.y = pack_bf16x2(acc[2 * j][2], acc[2 * j][3]);
      if (2 * j + 1 < NKT) {
        pa.z = pack_bf16x2(acc[2 * j + 1 < NKT ? 2 * j + 1 : 0][0], acc[2 * j + 1 < NKT ? 2 * j + 1 : 0][1]);
        pa.w = pack_bf16x2(acc[2 * j + 1 < NKT ? 2 * j + 1 : 0][2], acc[2 * j + 1 < NKT ? 2 * j + 1 : 0][3]);
      } else {
        pa.z = 0; pa.w = 0;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint16_t* vp = Vt + (size_t)(dt * 16 + c) * LDK + 2 * j * 16 + g * 4;
        const uint2 lo = *reinterpret_cast<const uint2*>(vp);
        uint2 hi = make_uint2(0, 0);
        if (2 * j + 1 < NKT) hi = *reinterpret_cast<const uint2*>(vp + 16);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(pa), as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y)),
                                                        o[dt], 0, 0, 0);
      }
    }
    // o[dt][r] = O[query g*4+r][d = dt*16 + c]; normalise, transpose through LDS, store whole rows
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float linv = 1.0f / __shfl(l, g * 4 + r, 64);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ot[(g * 4 + r) * OS + dt * 16 + c] = f32_to_bf16(o[dt][r] * linv);
    }
    // same wave wrote and reads: LDS ops of one wave complete in order, no barrier needed
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = (lane >> 3) + 8 * k, ch = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(ot + row * OS + ch * 8);
      const int q = qt * 16 + row;
      if (q < N) *reinterpret_cast<uint4*>(out + ((size_t)b * T + tok0 + q) * D + h * 64 + ch * 8) = v;
    }
    if (g == 0 && qrow < N) lse[((size_t)b * H + h) * T + tok0 + qrow] = m + __logf(l);
  }

  if constexpr (TEXT) return;
  // ---- CLS query partial over this frame's keys (key row 0 = the cls key itself: frame 0 only) -----------
  float s_loc[2];
  float mloc = -INFINITY;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = tid + 256 * k;
    float s = -INFINITY;
    if (j < nkeys && (j > 0 || f == 0)) {
      s = 0.f;
      const uint16_t* kp = Ks + j * KS;
#pragma unroll
      for (int d8 = 0; d8 < 8; ++d8) {
        float kv[8];
        Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(kp + d8 * 8), kv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qc[d8 * 8 + e], kv[e], s);
      }
    }
    s_loc[k] = s;
    mloc = fmaxf(mloc, s);
  }
  mloc = wave_max(mloc);
  if (lane == 0) red[256 + wave] = mloc;
  __syncthreads();
  const float M = fmaxf(fmaxf(red[256], red[257]), fmaxf(red[258], red[259]));
  float lsum = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = tid + 256 * k;
    const float p = s_loc[k] == -INFINITY ? 0.f : __expf(s_loc[k] - M);
    if (j < L::KROWS) sc[j] = p;
    lsum += p;
  }
  lsum = wave_sum(lsum);
  if (lane == 0) red[260 + wave] = lsum;
  __syncthreads();
  {
    // acc[d] = sum_j p_j V[j][d]; thread (d = lane, quarter = wave) walks a quarter of the key rows
    constexpr int QK = L::KROWS / 4;
    float a = 0.f;
    const uint16_t* vrow = Vt + (size_t)lane * LDK + wave * QK;
#pragma unroll 4
    for (int j = 0; j < QK; j += 4) {
      const uint2 v4 = *reinterpret_cast<const uint2*>(vrow + j);
      const float4 p4 = *reinterpret_cast<const float4*>(sc + wave * QK + j);
      a = fmaf(p4.x, __uint_as_float(v4.x << 16), a);
      a = fmaf(p4.y, __uint_as_float(v4.x & 0xffff0000u), a);
      a = fmaf(p4.z, __uint_as_float(v4.y << 16), a);
      a = fmaf(p4.w, __uint_as_float(v4.y & 0xffff0000u), a);
    }
    red[wave * 64 + lane] = a;
  }
  __syncthreads();
  if (tid < 64) {
    float* rec = cls_ws + (((size_t)b * H + h) * F + f) * CLS_REC;
    rec[2 + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
    if (tid == 0) {
      rec[0] = M;
      rec[1] = red[260] + red[261] + red[262] + red[263];
    }
  }
}

// merges the per-chunk partials of the CLS query: out[b,0,h,:] and lse[b,h,0]
__global__ __launch_bounds__(64) void cls_combine_kernel(const float* __restrict__ cls_ws, uint16_t* __restrict__ out,
                                                         float* __restrict__ lse, int nparts, int T, int H) {
  const int h = blockIdx.x % H, b = blockIdx.x / H, d = threadIdx.x;
  const float* rec = cls_ws + ((size_t)b * H + h) * nparts * CLS_REC;
  float M = -INFINITY;
  for (int p = 0; p < nparts; ++p) M = fmaxf(M, rec[p * CLS_REC]);
  float Lsum = 0.f, acc = 0.f;
  for (int p = 0; p < nparts; ++p) {
    const float w = __expf(rec[p * CLS_REC] - M);
    Lsum = fmaf(rec[p * CLS_REC + 1], w, Lsum);
    acc = fmaf(rec[p * CLS_REC + 2 + d], w, acc);
  }
  out[(size_t)b * T * H * 64 + h * 64 + d] = f32_to_bf16(acc / Lsum);
  if (d == 0) lse[((size_t)b * H + h) * T] = M + __logf(Lsum);
}

template <int NKT, bool TEXT = false>
int launch_space_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st) {
  using L = SpaceLds<NKT>;
  static_assert(L::total <= 160 * 1024, "LDS per CU");   // <= 80 KB (NKT <= 13) keeps 2 workgroups per CU
  if (L::total > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)space_fwd_kernel<NKT, TEXT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              L::total);
  hipLaunchKernelGGL((space_fwd_kernel<NKT, TEXT>), dim3((unsigned)(B * F * H)), dim3(256), L::total, st,
                     (const uint16_t*)qkv, (uint16_t*)out, lse, ws, F, N, H);
  LVL_CHECK_LAUNCH("space_fwd_mfma");
  if (TEXT) return LVL_OK;
  hipLaunchKernelGGL(cls_combine_kernel, dim3((unsigned)(B * H)), dim3(64), 0, st, ws, (uint16_t*)out, lse, F,
                     1 + F * N, H);
  LVL_CHECK_LAUNCH("cls_combine");
  return LVL_OK;
}

}  // namespace

void lvl_launch_cls_combine(const float* ws, void* out, float* lse, int B, int H, int nparts, int T, hipStream_t st) {
  hipLaunchKernelGGL(cls_combine_kernel, dim3((unsigned)(B * H)), dim3(64), 0, st, ws, (uint16_t*)out, lse, nparts, T, H);
}

bool lvl_space_mfma_supported(int F, int N) { return N + 1 <= 272 && N >= 1 && F <= 64; }

int lvl_space_mfma_fwd(const void* qkv, void* out, float* lse, float* ws, int B, int F, int N, int H, hipStream_t st) {
  const int nkeys = N + 1;
  if (nkeys <= 64) return launch_space_fwd<4>(qkv, out, lse, ws, B, F, N, H, st);
  if (nkeys <= 128) return launch_space_fwd<8>(qkv, out, lse, ws, B, F, N, H, st);
  if (nkeys <= 208) return launch_space_fwd<13>(qkv, out, lse, ws, B, F, N, H, st);
  if (nkeys <= 272) return launch_space_fwd<17>(qkv, out, lse, ws, B, F, N, H, st);
  return lvl_fail(LVL_ENOSYS, "space_mfma_fwd: %d keys per group exceeds the LDS-resident kernel", nkeys);
}

bool lvl_text_mfma_supported(int L) { return L >= 1 && L <= 272; }

int lvl_text_mfma_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, hipStream_t st) {
  if (L <= 64) return launch_space_fwd<4, true>(qkv, out, lse, nullptr, B, 1, L, H, st);
  if (L <= 128) return launch_space_fwd<8, true>(qkv, out, lse, nullptr, B, 1, L, H, st);
  if (L <= 208) return launch_space_fwd<13, true>(qkv, out, lse, nullptr, B, 1, L, H, st);
  if (L <= 272) return launch_space_fwd<17, true>(qkv, out, lse, nullptr, B, 1, L, H, st);
  return lvl_fail(LVL_ENOSYS, "text_mfma_fwd: context length %d exceeds the LDS-resident kernel", L);
}
