// Next-token sampling of the narrator (VCLM_HF.generate, narrator.py:118-143 + the warpers of :368-389), one kernel per
// decode step, gfx950. The reference does this with ~25 framework kernels per step on [captions, 50257] f32 tensors --
// softmax + entropy for the perplexity, temperature, top-k (topk), top-p (a full SORT of every row, cumsum, scatter),
// softmax, multinomial: at 640 captions (64 clips x 10 samples, the documented recipe) that is 3 ms per step, as much as
// the whole decoder. Here ONE workgroup per caption keeps the row in LDS as 16-bit order-preserving keys (the decoder's
// logits are bf16: 100 KB for GPT-2's 50257 entries) and works on it in place:
//   1. max, sum exp, sum exp * (l - max): the entropy of the UNWARPED distribution, or the cross entropy against a
//      target token (what generate() accumulates into the perplexity);
//   2. top-k: the k-th largest KEY by a two-level radix count (2048 + 32 buckets: bf16 keys have 16 bits, so two levels
//      are exact); entries below it are dropped (ties with the k-th value stay, as `scores < kth` does in transformers);
//   3. top-p on what is left, at temperature T: the boundary VALUE below which the ascending cumulative mass stays within
//      (1 - top_p) of the total -- transformers' rule `cumsum(sorted ascending) <= 1 - top_p` without sorting -- by
//      bisection over the 16 key bits, each round one pass over per-thread register copies of the weights and one block
//      sum; among entries that tie with the boundary value the first r in index order go (transformers drops r of them
//      too, which ones depends on its unstable sort), the largest entry always stays;
//   4. the draw: the kept entries in INDEX order, inverse CDF at uniform[row] * kept mass (block scan of per-thread
//      chunk sums, then one thread walks its chunk). torch.multinomial draws from the same distribution with another
//      mapping of random numbers to tokens; the uniforms come from torch's generator, so manual_seed still reproduces a run.
#include "common.h"

namespace {

constexpr int ST = 1024;            // threads per row
constexpr int L1B = 2048, L2B = 32; // radix levels over the 16-bit key: key >> 5, key & 31

__device__ __forceinline__ uint32_t key_of(uint16_t b) { return (b & 0x8000u) ? (uint16_t)~b : (uint16_t)(b | 0x8000u); }
__device__ __forceinline__ float val_of(uint32_t k) {
  const uint16_t b = (k & 0x8000u) ? (uint16_t)(k & 0x7fffu) : (uint16_t)~k;
  return bf16_to_f32(b);
}

struct Scratch {
  float red[2][ST / LVL_WAVE];
  float wave_tot[ST / LVL_WAVE];
  float bcast[8];
  unsigned ubcast[8];
};

__device__ __forceinline__ float block_sum(float v, Scratch& sc, int slot) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sc.red[slot][threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < ST / LVL_WAVE; ++i) t += sc.red[slot][i];
  __syncthreads();
  return t;
}

// exclusive prefix over the block in thread order; `total` = sum of all
__device__ __forceinline__ float block_exclusive_scan(float v, Scratch& sc, float& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) sc.wave_tot[wave] = inc;
  __syncthreads();
  float base = 0.f, tot = 0.f;
#pragma unroll
  for (int i = 0; i < ST / LVL_WAVE; ++i) {
    const float t = sc.wave_tot[i];
    if (i < wave) base += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

// First bucket b in WALK order (ascending index, or descending when DESC) whose running total -- the sum of the buckets
// walked before it plus its own content -- exceeds `thr` (STRICT) or reaches it (!STRICT); clamped to `limit` (the walk
// never goes past it). Returns b through sc.ubcast[0] and the total walked BEFORE b through sc.bcast[0]. Every thread
// owns two consecutive buckets of the walk; one block scan replaces a 2048-step serial walk.
template <bool DESC, bool STRICT, typename V>
__device__ __forceinline__ void find_boundary(const V* h, float thr, int limit, Scratch& sc) {
  const int t = threadIdx.x;
  const int b0 = DESC ? L1B - 1 - 2 * t : 2 * t, b1 = DESC ? b0 - 1 : b0 + 1;
  const float a = (float)h[b0], b = (float)h[b1];
  float tot;
  const float pre = block_exclusive_scan(a + b, sc, tot);
  if (t == 0) sc.ubcast[0] = (unsigned)(DESC ? L1B - 1 - limit : limit);     // walk position of the limit
  __syncthreads();
  const bool hit0 = STRICT ? pre + a > thr : pre + a >= thr;
  const bool hit1 = STRICT ? pre + a + b > thr : pre + a + b >= thr;
  if (hit0) atomicMin(&sc.ubcast[0], (unsigned)(2 * t));
  else if (hit1) atomicMin(&sc.ubcast[0], (unsigned)(2 * t + 1));
  __syncthreads();
  const unsigned pos = sc.ubcast[0];
  __syncthreads();
  if ((unsigned)t == pos / 2) {
    sc.bcast[0] = (pos & 1) ? pre + a : pre;
    sc.ubcast[0] = (unsigned)(DESC ? L1B - 1 - (int)pos : (int)pos);
  }
  __syncthreads();
}

__global__ __launch_bounds__(ST) void sample_kernel(const uint16_t* __restrict__ logits, int64_t row_stride, int V,
                                                    float inv_temp, int top_k, float top_p,
                                                    const float* __restrict__ uniform, const int64_t* __restrict__ target,
                                                    int64_t pad_id, int64_t* __restrict__ next_token,
                                                    float* __restrict__ nll, float* __restrict__ counted,
                                                    float* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint16_t* keys = reinterpret_cast<uint16_t*>(smem);                       // [Vp] (Vp = V rounded up to 8)
  const int Vp = (V + 7) & ~7;
  float* hmass = reinterpret_cast<float*>(smem + (size_t)Vp * 2);           // [L1B]
  unsigned* hcnt = reinterpret_cast<unsigned*>(hmass + L1B);                // [L1B]
  Scratch& sc = *reinterpret_cast<Scratch*>(hcnt + L1B);
  const int tid = threadIdx.x, row = blockIdx.x;
  const uint16_t* src = logits + (int64_t)row * row_stride;
  long long tick[6];                                         // phase boundaries (100 MHz wall clock), reported via dbg
  tick[0] = wall_clock64();

  // ---- 0. the row -> keys; maximum ------------------------------------------------------------------------------
  uint32_t kmax = 0;
  for (int i = tid * 8; i < Vp; i += ST * 8) {
    const uint4 raw = *reinterpret_cast<const uint4*>(src + i);          // row_stride % 8 == 0, padded columns exist
    const uint32_t wds[4] = {raw.x, raw.y, raw.z, raw.w};
    uint16_t k8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint16_t b = (uint16_t)(wds[j >> 1] >> ((j & 1) * 16));
      const uint32_t k = (i + j < V) ? key_of(b) : 0u;                   // columns >= V: the smallest key, never kept
      k8[j] = (uint16_t)k;
      kmax = k > kmax ? k : kmax;
    }
    *reinterpret_cast<uint4*>(keys + i) = *reinterpret_cast<const uint4*>(k8);
  }
  {
    float km = (float)kmax;                                               // 16-bit integers are exact in f32
    km = wave_max(km);
    if ((tid & 63) == 0) sc.red[0][tid >> 6] = km;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < ST / LVL_WAVE; ++i) t = fmaxf(t, sc.red[0][i]);
    kmax = (uint32_t)t;
    __syncthreads();
  }
  const float lmax = val_of(kmax);

  tick[1] = wall_clock64();
  // ---- 1. statistics of the unwarped distribution ------------------------------------------------------------------
  float s0 = 0.f, s1 = 0.f;
  for (int i = tid; i < V; i += ST) {
    const float d = val_of(keys[i]) - lmax;
    const float e = __expf(d);
    s0 += e;
    s1 = fmaf(e, d, s1);
  }
  s0 = block_sum(s0, sc, 0);
  s1 = block_sum(s1, sc, 1);
  if (tid == 0) {
    const float logz = __logf(s0);
    float out = logz - s1 / s0, cnt = 1.f;                                // entropy = log Z - E[l - max]
    if (target) {
      const int64_t tg = target[row];
      const bool ok = tg != pad_id && tg >= 0 && tg < V;
      out = ok ? logz - (val_of(keys[tg]) - lmax) : 0.f;                  // cross entropy, ignore_index = pad
      cnt = ok ? 1.f : 0.f;
    }
    nll[row] = out;
    counted[row] = cnt;
  }

  tick[2] = wall_clock64();
  // ---- 2. top-k: the k-th largest key -----------------------------------------------------------------------------------
  uint32_t kth = 1;                                                        // keep every real entry (padding has key 0)
  if (top_k == 1) {
    kth = kmax > 1 ? kmax : 1;                                            // greedy: the maximum (and its ties) is all that stays
  } else if (top_k > 0 && top_k < V) {
    for (int i = tid; i < L1B; i += ST) hcnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < V; i += ST) atomicAdd(&hcnt[keys[i] >> 5], 1u);
    __syncthreads();
    find_boundary<true, false>(hcnt, (float)top_k, 0, sc);                 // counts < 2^24: exact in f32
    if (tid == 0) sc.ubcast[1] = (unsigned)top_k - (unsigned)sc.bcast[0];  // rank wanted inside the bucket (>= 1)
    __syncthreads();
    const unsigned hb = sc.ubcast[0], want = sc.ubcast[1];
    __syncthreads();
    if (tid < L2B) hcnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < V; i += ST)
      if ((unsigned)(keys[i] >> 5) == hb) atomicAdd(&hcnt[keys[i] & 31], 1u);
    __syncthreads();
    if (tid == 0) {
      unsigned above = 0;
      int b = L2B - 1;
      for (; b > 0; --b) {
        if (above + hcnt[b] >= want) break;
        above += hcnt[b];
      }
      sc.ubcast[2] = (hb << 5) | (unsigned)b;
    }
    __syncthreads();
    kth = sc.ubcast[2] > 1 ? sc.ubcast[2] : 1;
    __syncthreads();
  }

  tick[3] = wall_clock64();
  // ---- 3. top-p: boundary key v*, r ties to drop ------------------------------------------------------------------------
  uint32_t vstar = 0;         // keys below vstar are dropped, keys above kept; among keys == vstar the first `rdrop` go
  unsigned rdrop = 0;
  if (top_p < 1.f) {
    // Every thread keeps the warped weights e_i = exp((l_i - max) / T) of its (strided) entries in registers; the
    // boundary is then found by BISECTION over the 16-bit key: M(v) = sum of e_i over kept keys <= v is one pass of
    // compares + adds over registers and one block sum -- 16 rounds at most. (A radix histogram of the mass needs
    // ds_add_f32 on a handful of hot buckets: measured 60 us per row; this is ~10.)
    constexpr int PER = 52;                                                // entries per thread: vocab <= 53248 (GPT-2: 50257)
    float e[PER];
    uint32_t kk[PER / 2];                                                  // their keys, two per register
    float z = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * ST;
      const uint32_t k = i < V ? keys[i] : 0u;                             // key 0 < kth: never counted
      e[u] = k >= kth ? __expf((val_of(k) - lmax) * inv_temp) : 0.f;
      z += e[u];
      if (u & 1) kk[u >> 1] |= k << 16; else kk[u >> 1] = k;
    }
    z = block_sum(z, sc, 0);
    const float thr = (1.f - top_p) * z;
    uint32_t blo = kth, bhi = kmax;                                        // M(kmax) = z > thr: the answer is <= kmax
    while (blo < bhi) {
      const uint32_t mid = (blo + bhi) >> 1;
      float part = 0.f;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const uint32_t k = (u & 1) ? kk[u >> 1] >> 16 : kk[u >> 1] & 0xffffu;
        part += k <= mid ? e[u] : 0.f;                                     // e is 0 for dropped / absent entries
      }
      part = block_sum(part, sc, 0);
      if (part > thr) bhi = mid; else blo = mid + 1;
    }
    const uint32_t v = blo;                                                // smallest key whose cumulative mass exceeds thr
    float below = 0.f, cntf = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const uint32_t k = (u & 1) ? kk[u >> 1] >> 16 : kk[u >> 1] & 0xffffu;
      below += k < v ? e[u] : 0.f;
      cntf += (k == v) ? 1.f : 0.f;                                        // v >= kth >= 1 > the padding key 0
    }
    below = block_sum(below, sc, 0);
    cntf = block_sum(cntf, sc, 1);
    const float p = __expf((val_of(v) - lmax) * inv_temp);
    const unsigned cnt = (unsigned)cntf;
    float fr = floorf((thr - below) / p);
    fr = fr < 0.f ? 0.f : fr;
    unsigned r = fr >= (float)cnt ? cnt : (unsigned)fr;
    if (v == kmax && r >= cnt) r = cnt > 0 ? cnt - 1 : 0;                  // min_tokens_to_keep = 1: the largest stays
    vstar = v;
    rdrop = r;
  }
  const uint32_t lo = vstar > kth ? vstar : kth;                            // keys below `lo` are out

  tick[4] = wall_clock64();
  // ---- 4. the draw --------------------------------------------------------------------------------------------------
  const int chunk = (V + ST - 1) / ST;
  const int c0 = tid * chunk, c1 = c0 + chunk < V ? c0 + chunk : V;
  float ties = 0.f;
  if (rdrop > 0)
    for (int i = c0; i < c1; ++i) ties += (keys[i] == vstar) ? 1.f : 0.f;
  float tie_total = 0.f;
  float tie_before = rdrop > 0 ? block_exclusive_scan(ties, sc, tie_total) : 0.f;
  float mass = 0.f;
  {
    float rank = tie_before;
    for (int i = c0; i < c1; ++i) {
      const uint32_t k = keys[i];
      bool keep = k >= lo;
      if (rdrop > 0 && k == vstar) {
        keep = keep && rank >= (float)rdrop;
        rank += 1.f;
      }
      if (keep) mass += __expf((val_of(k) - lmax) * inv_temp);
    }
  }
  float zk = 0.f;
  const float before = block_exclusive_scan(mass, sc, zk);
  const float want = uniform[row] * zk;
  if (tid == 0) sc.ubcast[3] = 0xffffffffu;
  __syncthreads();
  // top_k = 1 is greedy decoding: the FIRST maximum, like torch.argmax (entries tied with it are not drawn among)
  if (top_k != 1 && mass > 0.f && want >= before && want < before + mass) {
    float run = before, rank = tie_before;
    int pick = -1;
    for (int i = c0; i < c1; ++i) {
      const uint32_t k = keys[i];
      bool keep = k >= lo;
      if (rdrop > 0 && k == vstar) {
        keep = keep && rank >= (float)rdrop;
        rank += 1.f;
      }
      if (keep) {
        pick = i;                                                          // the last kept entry catches rounding at the end
        run += __expf((val_of(k) - lmax) * inv_temp);
        if (want < run) break;
      }
    }
    sc.ubcast[3] = (unsigned)pick;
  }
  __syncthreads();
  if (sc.ubcast[3] == 0xffffffffu) {                                        // greedy, or uniform * zk landed on / beyond the
    for (int i = c0; i < c1; ++i)                                          // total: the (first) most probable token
      if (keys[i] == kmax) atomicMin(&sc.ubcast[3], (unsigned)i);
  }
  __syncthreads();
  if (tid == 0) {
    next_token[row] = (int64_t)sc.ubcast[3];
    if (dbg) {
      tick[5] = wall_clock64();
      dbg[row * 12 + 0] = val_of(lo);
      dbg[row * 12 + 1] = (float)rdrop;
      dbg[row * 12 + 2] = zk;
      dbg[row * 12 + 3] = val_of(vstar);
#pragma unroll
      for (int i = 0; i < 5; ++i) dbg[row * 12 + 4 + i] = (float)(tick[i + 1] - tick[i]) * 0.01f;   // us per phase
      dbg[row * 12 + 9] = (float)(tick[0] % 100000000ll) * 0.01f;                                   // start time, us
    }
  }
}

size_t sample_lds_bytes(int V) {
  const size_t Vp = (size_t)((V + 7) & ~7);
  return Vp * 2 + (size_t)L1B * 8 + sizeof(Scratch) + 64;
}

}  // namespace

extern "C" int lvl_sample_max_vocab() {
  const int lds = (160 * 1024 - (int)(L1B * 8 + sizeof(Scratch) + 64)) / 2 - 8;
  return lds < 52 * ST ? lds : 52 * ST;      // 52 = entries per thread the top-p pass keeps in registers
}

extern "C" int lvl_sample_next_token(const void* logits, int64_t row_stride, int rows, int vocab, float temperature,
                                     int top_k, float top_p, const float* uniform, const int64_t* target, int64_t pad_id,
                                     int64_t* next_token, float* nll, float* counted, float* dbg, void* stream) {
  LVL_REQUIRE(rows == 0 || (logits && uniform && next_token && nll && counted), "sample_next_token: null pointer");
  LVL_REQUIRE(rows >= 0 && vocab > 0 && temperature > 0.f && top_k >= 0 && top_p > 0.f,
              "sample_next_token: bad arguments rows=%d vocab=%d temperature=%g top_k=%d top_p=%g", rows, vocab,
              (double)temperature, top_k, (double)top_p);
  if (vocab > lvl_sample_max_vocab())
    return lvl_fail(LVL_ENOSYS, "sample_next_token: a row of %d logits does not fit one workgroup's LDS", vocab);
  LVL_REQUIRE(row_stride % 8 == 0 && row_stride >= ((vocab + 7) & ~7) && lvl_aligned16(logits),
              "sample_next_token: rows must be 16-byte aligned and padded to a multiple of 8 columns (stride %lld)",
              (long long)row_stride);
  if (rows == 0) return LVL_OK;
  if (int rc = lvl_allow_lds<sample_kernel>()) return rc;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)rows), dim3(ST), sample_lds_bytes(vocab), (hipStream_t)stream,
                     (const uint16_t*)logits, row_stride, vocab, 1.f / temperature, top_k, top_p, uniform, target, pad_id,
                     next_token, nll, counted, dbg);
  LVL_CHECK_LAUNCH("sample_next_token");
  return LVL_OK;
}
