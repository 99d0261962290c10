// Weight gradient of the token-major Linear layers: dW[N,K] = dY[M,N]^T X[M,K] (+ dbias[N] = column sums of dY),
// bf16 operands, f32 accumulation, gfx950 MFMA.
//
// Shape of the problem on this path: M = B*T ~ 2e5 rows, N,K in {768, 2304, 3072}: a tiny output contracted over
// a huge row count, with BOTH operands stored row-major over the contraction index (the layout a library GEMM
// likes least: every MFMA fragment needs a transpose). Here:
//   * the output is cut into tiles, the rows into S splits; one workgroup of WN*WK waves per (tile, split), all of
//     them resident at once. A wave owns TA x TB MFMA tiles in registers: 6x6 (96x96; workgroup tile 384x192,
//     192x384, 288x192 ... for the 768-family widths) or 8x4 (128x64; workgroup tile 256x256, 256x128, 128x256 for
//     the 512- and 1024-family widths);
//   * blockIdx -> (split, tile) is XCD-aware: workgroup i runs on XCD i%8, and XCD x is given a contiguous range
//     of the split-major (split, tile) pairs, so the ~32 workgroups of an XCD stream the SAME dY/X rows through
//     that XCD's L2 (each row is fetched from HBM about once instead of once per tile);
//   * rows are staged 32 at a time as row-major LDS images (row stride padded by 32 B -> an odd number of 32-B
//     bank groups, conflict-free for the reads below; SQ_LDS_BANK_CONFLICT = 0) by LDS-DMA
//     (global_load_lds_dwordx4, no VGPR round trip) into a 4-deep ring: fills run 2-3 steps ahead, one bare
//     s_barrier per step placed mid-step so that neither fill nor LDS latency separates two steps' MFMAs;
//   * both MFMA operands are read with ds_read_b64_tr_b16 (the gfx950 LDS transpose read): lane c of a 16-lane
//     group receives 4 consecutive ROWS at column c, two reads give the 8 contraction elements of a
//     v_mfma_f32_16x16x32_bf16 operand. A and B use the same row permutation, so no data is ever transposed;
//   * dbias rides along: v_dot2_f32_bf16 of the A fragments with (1,1).
// Partial tiles [S,N,K] f32 are reduced by wgrad_reduce_kernel (deterministic, no atomics).
//
// Robustness against a taken compute unit (`sched`, optional): every (tile, split) unit is cut into C row CHUNKS that
// are handed out by a per-unit device counter. A workgroup claims the chunks of ITS OWN unit first -- one returning
// atomic per chunk boundary; while the replies are consecutive the row stream and the fragment pipeline run on across
// the boundary -- and when its unit is exhausted it takes unclaimed chunks of the other splits of the SAME tile into its own accumulators
// (its partial slab then simply holds more rows; the slab sum is unchanged). A workgroup whose CU is held by another
// kernel (an RCCL channel) therefore delays the launch by its tile-mates' share of its rows, not by a second round of
// the whole kernel; when it finally starts it finds its chunks gone and writes a zero slab. When nobody steals, each
// slab holds exactly the rows of the static plan, in the same order, and the result is bit-identical to the static
// schedule. That is the common case with every CU available, NOT a guarantee: a workgroup that finishes its unit takes
// chunks from a tile-mate that is merely slower by more than one chunk (XCD / HBM jitter), the slabs then hold other
// row sets and the f32 summation order of dW depends on timing (exact on integer operands, last-bit differences
// otherwise). Bit-reproducible runs: LAVILA_DYNAMIC_TILES=0 (static plan; the default on a single GPU).
#include "common.h"

int lvl_debug_late_mod();

// the LDS-DMA fills set M0 inside inline asm and say so in the clobber list; this kernel has no other M0 user
#pragma clang diagnostic ignored "-Winline-asm"

typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 wg_bf16x2;
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;
typedef __attribute__((ext_vector_type(4))) short wg_s16x4;

namespace {

constexpr int MS = 32;          // rows per step (one MFMA contraction)

__device__ __forceinline__ uint2 tr_read(const uint16_t* p) {
  const wg_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ wg_f32x4 mfma16(uint4 a, uint4 b, wg_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, a), __builtin_bit_cast(wg_bf16x8, b),
                                                 c, 0, 0, 0);
}

// Row plan of a launch (host-computed, no divisions in the kernel): the M/32 full row blocks are dealt to the S splits
// as `base` blocks each, the first `rem` splits one more; the dynamic schedule cuts a unit into chunks of L blocks
// (L even), nch0 / nch1 of them for a unit of base / base + 1 blocks.
struct RowPlan { int base, rem, L, nch0, nch1, late_mod; };

template <int WN, int WK, int TA, int TB>
struct Geo {
  static constexpr int NW = WN * WK, NT = 64 * NW;
  static constexpr int TN = 16 * TA * WN, TK = 16 * TB * WK;     // workgroup tile (rows of dW x columns of dW)
  static constexpr int SA = TN + 16, SB = TK + 16;            // image row strides in elements (+32 B)
  // LDS-DMA plan: one global_load_lds_dwordx4 fills 64 consecutive 16-B chunks (1 KiB) of a stage. An image of
  // MS rows x (stride/8) chunks is exactly IA (IB) such fills; the pad chunks of a row carry don't-care data.
  static constexpr int IA = MS * (SA / 8) / 64, IB = MS * (SB / 8) / 64;      // = TA*WN + 1, TB*WK + 1
  static constexpr int NI = (IA + IB + NW - 1) / NW;                          // fills per wave per step
  static constexpr int STAGE = MS * (SA + SB) + (NI * NW - IA - IB) * 512;    // elements per stage (+ dump area)
  static constexpr int NSTAGE = (4 * STAGE * 2 <= 160 * 1024) ? 4 : 3;   // ring depth: fills run NSTAGE-1 steps ahead
  static_assert(MS * (SA / 8) % 64 == 0 && MS * (SB / 8) % 64 == 0, "images must be whole 1-KiB fills");
};

template <int WN, int WK, int TA, int TB, bool BIAS>
__global__ __launch_bounds__(64 * WN * WK) void wgrad_kernel(const uint16_t* __restrict__ dy,
                                                             const uint16_t* __restrict__ x, float* __restrict__ part,
                                                             float* __restrict__ bpart, int64_t M, int N, int K,
                                                             int tiles_k, int ntiles, int S, RowPlan rp,
                                                             unsigned* __restrict__ sched) {
  using G = Geo<WN, WK, TA, TB>;
  static_assert(TA % 2 == 0, "the mid-step barrier splits the A tiles in two halves");
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];     // [NSTAGE][A image | B image | dump]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wk = wave / WN;
  // XCD-aware decode: consecutive block ids alternate XCDs, so XCD x = bid % 8 is given the contiguous range
  // [x*P/8, (x+1)*P/8) of the split-major (split, tile) pairs, P = ntiles*S: the workgroups that stream the same
  // rows sit behind the same L2
  const int bid = blockIdx.x;
  const int pair = (bid & 7) * ((ntiles * S) >> 3) + (bid >> 3);
  const int split = pair / ntiles, tile = pair % ntiles;
  const int n0 = (tile / tiles_k) * G::TN, k0 = (tile % tiles_k) * G::TK;
  const int64_t steps_total = M / MS;      // full row blocks; the < 32 tail rows are added by wgrad_reduce_kernel
  // (the dbias instantiations keep the static plan: they are not on the training path -- the bias gradients come from
  // the LayerNorm / GEMM epilogues -- and their register budget leaves no room for the claim state)
  const bool dyn = !BIAS && sched != nullptr;

  // ---- staging plan (LDS-DMA, no VGPR round trip) ---------------------------------------------------------------
  // Fill f (0 .. NI*NW-1) of a step is issued by wave f % NW; lane l of the fill lands at stage byte f*1024 + l*16.
  // Fills 0..IA-1 tile the A image, IA..IA+IB-1 the B image, the rest (so that every wave issues exactly NI fills
  // and one s_waitcnt immediate fits all) land in a dump area behind the images. Pad chunks and dump chunks read
  // chunk 0 of their row. Per-lane source pointers advance by one row block per step.
  const uint16_t* src[G::NI];
  int64_t src_step[G::NI];
  int dst_off[G::NI];             // element offset of the fill inside a stage (wave-uniform)
  // point the fill sources at row block `rb` (the first step of a stream)
  auto set_sources = [&](int64_t rb) {
#pragma unroll
    for (int q = 0; q < G::NI; ++q) {
      const int f = wave + q * G::NW;
      dst_off[q] = f * 512;
      const int chunk = f * 64 + lane;
      if (f < G::IA) {
        const int row = chunk / (G::SA / 8), c8 = chunk % (G::SA / 8);
        src[q] = dy + (rb * MS + row) * (int64_t)N + n0 + (c8 < G::TN / 8 ? c8 : 0) * 8;
        src_step[q] = (int64_t)MS * N;
      } else {
        const int cb = f < G::IA + G::IB ? chunk - G::IA * 64 : lane;
        const int row = cb / (G::SB / 8), c8 = cb % (G::SB / 8);
        src[q] = x + (rb * MS + row) * (int64_t)K + k0 + (c8 < G::TK / 8 ? c8 : 0) * 8;
        src_step[q] = (int64_t)MS * K;
      }
    }
  };
  // Fills are issued through inline asm: the compiler's LDS-DMA alias tracking would otherwise put s_waitcnt
  // vmcnt(0) in front of every LDS read and drain the run-ahead fills. Steps at or beyond `last_step` (run-ahead
  // past the end of the matrix) re-read the last full row block.
  int last_step = 0;              // steps_total - 1 - (first row block of the stream)
  int issued = 0;                 // steps issued so far in this stream; src[] points at step min(issued, last_step)
  auto issue_loads = [&](int stage) {
    const uint32_t lds_base = (uint32_t)(uintptr_t)(smem + stage * G::STAGE) ;
#pragma unroll
    for (int q = 0; q < G::NI; ++q) {
      const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)dst_off[q] * 2u);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src[q]) : "memory", "m0");
      if (issued < last_step) src[q] += src_step[q];
    }
    ++issued;
  };

  // ---- chunk claims (dynamic schedule only) ---------------------------------------------------------------------
  // A claim is ONE asm block on wave 0, lane 0 (EXEC narrowed inside it): returning atomic add -> s_waitcnt vmcnt(0) ->
  // ds_write of the reply into the mailbox. The reply never lives in a compiler-visible register across the wait (a
  // parked reply would be at the mercy of live-range splitting). The wait also drains this wave's run-ahead fills,
  // about 1 us once per chunk of >= 24 steps: < 1 % of the kernel. The mailbox is the 32 pad bytes behind row 0 of a
  // stage's dY image: never read by a fragment load, rewritten (with don't-care data) only by that stage's next fill.
  auto mbox_addr = [&](int stage) { return (uint32_t)(uintptr_t)(smem + stage * G::STAGE + G::TN); };
  uint64_t exec_save;
  uint32_t reply;
  auto claim = [&](unsigned* ctr, int stage) {          // mailbox <- old value of *ctr; *ctr += 1
    if (wave == 0)
      asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\t"
                   "s_waitcnt vmcnt(0)\n\tds_write_b32 %5, %0\n\ts_mov_b64 exec, %1"
                   : "=&v"(reply), "=&s"(exec_save) : "v"(0u), "v"(1u), "s"(ctr), "v"(mbox_addr(stage)) : "memory");
  };
  auto peek = [&](unsigned* ctr, int stage) {           // mailbox <- *ctr (agent-scope load)
    if (wave == 0)
      asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_load_dword %0, %2, %3 sc1\n\t"
                   "s_waitcnt vmcnt(0)\n\tds_write_b32 %4, %0\n\ts_mov_b64 exec, %1"
                   : "=&v"(reply), "=&s"(exec_save) : "v"(0u), "s"(ctr), "v"(mbox_addr(stage)) : "memory");
  };
  auto count_one = [&](unsigned* ctr) {                 // fire-and-forget atomic add of 1
    if (wave == 0)
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %1, %2, %3\n\ts_mov_b64 exec, %0"
                   : "=&s"(exec_save) : "v"(0u), "v"(1u), "s"(ctr) : "memory");
  };
  // mailbox -> every wave; two barriers: write -> read, read -> anything that may overwrite the mailbox
  auto collect = [&](int stage) -> int {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (asm LDS read: a volatile C++ read becomes a flat load with a vmcnt(0) drain of every wave's run-ahead fills)
    uint32_t raw;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(raw) : "v"(mbox_addr(stage)) : "memory");
    const int v = __builtin_amdgcn_readfirstlane(raw);
    __builtin_amdgcn_s_barrier();
    return v;
  };

  // ---- per-lane fragment offsets -----------------------------------------------------------------------------
  // tr read: lane (g = lane>>4, mm = lane&15) points at row rb + g*4 + mm/4, columns col0 + 4*(mm%4)..+3 and
  // receives rows rb + g*4 .. +3 at column col0 + mm. rb = 0 and 16 -> contraction order (g*4+e | 16+g*4+e).
  const int g = lane >> 4, mm = lane & 15;
  const int rsub = g * 4 + (mm >> 2), csub = (mm & 3) * 4;
  const int offA = rsub * G::SA + wn * (16 * TA) + csub;                  // + i*16 (+ 16*SA for the second half)
  const int offB = MS * G::SA + rsub * G::SB + wk * (16 * TB) + csub;

  wg_f32x4 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = wg_f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[TA];
#pragma unroll
  for (int i = 0; i < TA; ++i) bsum[i] = 0.f;
  const wg_bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};

  // NSTAGE-deep ring with the barrier in the MIDDLE of a step. Step s multiplies stage s%NSTAGE in two halves of
  // TA/2 A-tiles each. Between the halves: wait until this wave's fills of step s+1 have landed (vmcnt counts them
  // in order), s_barrier (=> step s+1 is complete for everybody, and everybody is done with step s-1), issue the
  // fills of step s+NSTAGE-1 into the stage step s-1 used, then read the B fragments of step s+1 into a second
  // register set while the second half's MFMAs run. No LDS latency and no fill latency sits between the last MFMA
  // of one step and the first of the next. The barrier is the bare s_barrier: a fence would drain the run-ahead.
  static_assert(G::NSTAGE == 4, "the mid-step schedule is written for a 4-deep ring");
  auto read_b = [&](const uint16_t* img, uint4 (&bf)[TB]) {
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const uint2 lo = tr_read(img + offB + j * 16), hi = tr_read(img + offB + j * 16 + 16 * G::SB);
      bf[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  };
  auto read_a = [&](const uint16_t* img, int i) {
    const uint2 lo = tr_read(img + offA + i * 16), hi = tr_read(img + offA + i * 16 + 16 * G::SA);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  };
  auto bias_dot = [&](int i, const uint4& af) {
    if (BIAS && (i % WK) == wk) {       // dbias: this wave's share of the n-tiles
      float b = bsum[i];
      b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wg_bf16x2, af.x), ones, b, false);
      b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wg_bf16x2, af.y), ones, b, false);
      b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wg_bf16x2, af.z), ones, b, false);
      b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wg_bf16x2, af.w), ones, b, false);
      bsum[i] = b;
    }
  };
  uint4 bf[TB], bf_next[TB], af_next;
  int stage = 0;
  // one step; `cur` holds this step's B fragments, `nxt` receives the next step's (ping-pong: no register copies)
  auto do_step = [&](uint4 (&cur)[TB], uint4 (&nxt)[TB]) {
    const uint16_t* img = smem + stage * G::STAGE;
    const int nstage = stage == G::NSTAGE - 1 ? 0 : stage + 1;
    const uint16_t* img_next = smem + nstage * G::STAGE;
#pragma unroll
    for (int i = 0; i < TA / 2; ++i) {
      const uint4 af = af_next;
      af_next = read_a(img, i + 1);
      bias_dot(i, af);
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma16(af, cur[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NI) : "memory");      // fills of step+1 landed (step+2 may be in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_loads(stage == 0 ? G::NSTAGE - 1 : stage - 1);               // step+3 -> the stage step-1 used
    read_b(img_next, nxt);
#pragma unroll
    for (int i = TA / 2; i < TA; ++i) {
      const uint4 af = af_next;
      af_next = i + 1 < TA ? read_a(img, i + 1) : read_a(img_next, 0);
      bias_dot(i, af);
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma16(af, cur[j], acc[i][j]);
    }
    stage = nstage;
  };

  // Units visited: the own (tile, split) first, then -- dynamic schedule only -- the other splits of the same tile.
  unsigned* const taken = dyn ? sched + ntiles * S + tile : nullptr;      // chunks of this tile consumed so far
  // (test hook, lvl_debug_late_workgroups: a "late" workgroup visits nothing and writes a zero slab)
  const int visits = dyn ? ((rp.late_mod > 0 && bid % rp.late_mod == 1) ? 0 : S) : 1;
  for (int d = 0; d < visits; ++d) {
    int sp = split + d;
    if (sp >= S) sp -= S;
    const int ulen = rp.base + (sp < rp.rem ? 1 : 0);                               // row blocks of the unit
    const int64_t ub = (int64_t)sp * rp.base + (sp < rp.rem ? sp : rp.rem), ue = ub + ulen;
    int L = ulen, nchunks = ulen > 0 ? 1 : 0, cur = 0;
    unsigned* uctr = nullptr;
    if (dyn) {
      L = rp.L;      // even: the fragment ping-pong is back in phase at every boundary the stream may run across
      nchunks = sp < rp.rem ? rp.nch1 : rp.nch0;
      uctr = sched + sp * ntiles + tile;
      if (d > 0) {
        // steal only if the tile still has unconsumed chunks (one load; stale by at most the claims in flight)
        const int total = rp.rem * rp.nch1 + (S - rp.rem) * rp.nch0;
        peek(taken, 0);
        if (collect(0) >= total) break;
      }
      claim(uctr, 0);
      cur = collect(0);
    }
    while (cur < nchunks) {
      // ---- one stream: chunk `cur` and, while the claims keep returning the next index, the chunks behind it ----
      if (dyn) count_one(taken);
      int64_t pos = ub + (int64_t)cur * L;
      set_sources(pos);
      last_step = (int)(steps_total - 1 - pos);
      issued = 0;
      stage = 0;
      issue_loads(0);
      issue_loads(1);
      issue_loads(2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::NI) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      read_b(smem, bf);
      af_next = read_a(smem, 0);
      int next = nchunks;            // where the next stream of this unit starts (>= nchunks: none)
      for (;;) {
        const int len = (int)(ue - pos < L ? ue - pos : L);
        const bool more = dyn && cur + 1 < nchunks;
        int step = 0;
        for (; step + 1 < len; step += 2) {
          do_step(bf, bf_next);
          do_step(bf_next, bf);
        }
        if (step < len) do_step(bf, bf_next);          // odd length: only a unit's last chunk
        if (!more) break;
        // every wave has finished the step that used stage `done`: its pad is free until the next step refills it
        const int done = stage == 0 ? G::NSTAGE - 1 : stage - 1;
        claim(uctr, done);
        next = collect(done);
        if (next != cur + 1 || (len & 1)) break;       // a tile-mate took chunks of this unit: stop, restart at `next`
        count_one(taken);
        cur = next;
        next = nchunks;
        pos += L;
      }
      // the run-ahead fills must have landed (and everybody must be done reading) before the stages are reused
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = next;
    }
  }

  // ---- epilogue: partial tile of this split -------------------------------------------------------------------
  float* out = part + ((size_t)split * N + n0 + wn * (16 * TA)) * K + k0 + wk * (16 * TB);
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)(i * 16 + g * 4 + r) * K + j * 16 + mm] = acc[i][j][r];
  if (BIAS && k0 == 0) {
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      if ((i % WK) == wk) {
        float b = bsum[i];
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (g == 0) bpart[(size_t)split * N + n0 + wn * (16 * TA) + i * 16 + mm] = b;
      }
    }
  }
  if (dyn && wave == 0) {
    // sign-off: this workgroup's counter traffic is complete (every stream ended with vmcnt(0)); the last workgroup
    // out leaves the counter block zeroed for the next launch that is handed the same block
    const int nctr = ntiles * S + ntiles;          // unit counters | per-tile consumed counts | [nctr] = sign-offs
    unsigned gone = 0;
    if (lane == 0) gone = __hip_atomic_fetch_add(sched + nctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gone = __builtin_amdgcn_readfirstlane(gone);
    if (gone == gridDim.x - 1)
      for (int q = lane; q <= nctr; q += 64) __hip_atomic_store(sched + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// dw[e] = sum_s part[s][e] (+ the < 32 tail rows m >= M_main, which the tiled kernel skips), db[n] likewise
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part,
                                                           const float* __restrict__ bpart,
                                                           const uint16_t* __restrict__ dy,
                                                           const uint16_t* __restrict__ x, float* __restrict__ dw,
                                                           float* __restrict__ db, int64_t M_main, int64_t M, int N,
                                                           int K, int S) {
  const int64_t NK = (int64_t)N * K;
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v * 4 < NK) {
    // fixed summation order s = 0..S-1; the loads of 4 slabs are in flight together
    float4 a = *reinterpret_cast<const float4*>(part + v * 4);
    int s = 1;
    for (; s + 3 < S; s += 4) {
      const float4 b0 = *reinterpret_cast<const float4*>(part + (size_t)s * NK + v * 4);
      const float4 b1 = *reinterpret_cast<const float4*>(part + (size_t)(s + 1) * NK + v * 4);
      const float4 b2 = *reinterpret_cast<const float4*>(part + (size_t)(s + 2) * NK + v * 4);
      const float4 b3 = *reinterpret_cast<const float4*>(part + (size_t)(s + 3) * NK + v * 4);
      a.x = (((a.x + b0.x) + b1.x) + b2.x) + b3.x;
      a.y = (((a.y + b0.y) + b1.y) + b2.y) + b3.y;
      a.z = (((a.z + b0.z) + b1.z) + b2.z) + b3.z;
      a.w = (((a.w + b0.w) + b1.w) + b2.w) + b3.w;
    }
    for (; s < S; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(part + (size_t)s * NK + v * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const int n = (int)((v * 4) / K), k = (int)((v * 4) % K);
    for (int64_t m = M_main; m < M; ++m) {
      const float d = bf16_to_f32(dy[m * N + n]);
      const uint2 xr = *reinterpret_cast<const uint2*>(x + m * K + k);
      a.x = fmaf(d, __uint_as_float(xr.x << 16), a.x);
      a.y = fmaf(d, __uint_as_float(xr.x & 0xffff0000u), a.y);
      a.z = fmaf(d, __uint_as_float(xr.y << 16), a.z);
      a.w = fmaf(d, __uint_as_float(xr.y & 0xffff0000u), a.w);
    }
    *reinterpret_cast<float4*>(dw + v * 4) = a;
  }
  if (db != nullptr && v < N) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += bpart[(size_t)s * N + v];
    for (int64_t m = M_main; m < M; ++m) a += bf16_to_f32(dy[m * N + v]);
    db[v] = a;
  }
}

struct Plan { int cfg, tiles_k, ntiles, S; bool ok; };

// workgroup shapes: {WN, WK, TA, TB}; tile = (16*TA*WN) x (16*TB*WK)
constexpr int kCfg[][4] = {{4, 2, 6, 6}, {2, 4, 6, 6}, {3, 2, 6, 6}, {2, 3, 6, 6}, {2, 2, 6, 6},
                           {2, 4, 8, 4}, {2, 2, 8, 4}, {1, 4, 8, 4}};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

// M > 0: the row count of the launch -- a unit keeps at least ~48 steps (1536 rows) so that a short problem (the text
// tower: 8192 rows) is not cut into 250 units whose 256-KiB partial slabs cost more than their MFMAs and which would
// hold every CU while the other tower's kernels wait. M = 0: the largest plan (workspace sizing).
Plan make_plan(int N, int K, int64_t M = 0) {
  Plan best{};
  int64_t best_score = 0;
  for (int ci = 0; ci < kNumCfg; ++ci) {
    const int* c = kCfg[ci];
    const int tn = 16 * c[2] * c[0], tk = 16 * c[3] * c[1];
    if (N % tn || K % tk) continue;
    const int ntiles = (N / tn) * (K / tk);
    const int cus = lvl_persistent_cus();
    if (ntiles > cus) continue;
    // one resident workgroup per CU: S * ntiles <= CUs and a multiple of 8 (XCD mapping)
    const int smax = cus / ntiles;
    int S = smax;
    if (M > 0) {
      const int64_t cap = (M / MS) / 48;
      if (S > cap) S = cap < 1 ? 1 : (int)cap;
    }
    const int want = S;
    while (S > 1 && (ntiles * S) % 8) --S;
    if ((ntiles * S) % 8) {                  // nothing at or below the cap maps onto whole XCD rounds: go up instead
      S = want;
      while (S <= smax && (ntiles * S) % 8) ++S;
      if (S > smax) continue;
    }
    // score = busy SIMD slots x tile area: 6 waves load the 4 SIMDs 2:2:1:1, 4 waves leave every SIMD one wave
    const int waves = c[0] * c[1];
    const int64_t eff = waves % 4 == 0 ? (waves >= 8 ? 4 : 3) : 3;
    const int64_t score = (int64_t)ntiles * S * eff * 1000 + (int64_t)tn * tk / 64;
    if (score > best_score) {
      best_score = score;
      best = Plan{ci, K / tk, ntiles, S, true};
    }
  }
  return best;
}

// chunks per (tile, split) unit of the dynamic schedule: at least ~48 steps (1536 rows, ~30 us) each, at most 4
RowPlan row_plan(int64_t M, int S) {
  const int64_t steps = M / MS;
  RowPlan rp;
  rp.base = (int)(steps / S);
  rp.rem = (int)(steps % S);
  int c = rp.base / 48;
  c = c < 1 ? 1 : (c > 4 ? 4 : c);
  rp.L = ((rp.base + 1 + c - 1) / c + 1) & ~1;
  if (rp.L < 2) rp.L = 2;
  rp.nch0 = (rp.base + rp.L - 1) / rp.L;
  rp.nch1 = (rp.base + 1 + rp.L - 1) / rp.L;
  rp.late_mod = lvl_debug_late_mod();
  return rp;
}

template <int WN, int WK, int TA, int TB>
int launch(const Plan& p, const void* dy, const void* x, float* part, float* bpart, int64_t M, int N, int K,
           unsigned* sched, hipStream_t st) {
  const RowPlan rp = row_plan(M, p.S);
  using G = Geo<WN, WK, TA, TB>;
  const size_t shmem = (size_t)G::NSTAGE * G::STAGE * sizeof(uint16_t);
  const dim3 grid((unsigned)(p.ntiles * p.S)), block(G::NT);
  if (bpart != nullptr) {
    if (int rc = lvl_allow_lds<wgrad_kernel<WN, WK, TA, TB, true>>()) return rc;
    hipLaunchKernelGGL((wgrad_kernel<WN, WK, TA, TB, true>), grid, block, shmem, st, (const uint16_t*)dy,
                       (const uint16_t*)x, part, bpart, M, N, K, p.tiles_k, p.ntiles, p.S, rp, sched);
  } else {
    if (int rc = lvl_allow_lds<wgrad_kernel<WN, WK, TA, TB, false>>()) return rc;
    hipLaunchKernelGGL((wgrad_kernel<WN, WK, TA, TB, false>), grid, block, shmem, st, (const uint16_t*)dy,
                       (const uint16_t*)x, part, bpart, M, N, K, p.tiles_k, p.ntiles, p.S, rp, sched);
  }
  LVL_CHECK_LAUNCH("linear_wgrad");
  return LVL_OK;
}

}  // namespace

int64_t lvl_wgrad_workspace_floats(int64_t N, int64_t K) {
  const Plan p = make_plan((int)N, (int)K);
  if (!p.ok) return -1;
  return (int64_t)p.S * N * K + (int64_t)p.S * N;
}

extern "C" int lvl_linear_wgrad(const void* dy, const void* x, float* dw, float* dbias, float* ws, uint32_t* sched,
                                int64_t M, int N, int K, int dtype, void* stream) {
  LVL_REQUIRE(dy && x && dw && ws, "linear_wgrad: null pointer");
  LVL_REQUIRE(dtype == LVL_BF16, "linear_wgrad: bf16 operands only (dtype=%d)", dtype);
  LVL_REQUIRE(M > 0 && N > 0 && K > 0, "linear_wgrad: empty problem");
  LVL_REQUIRE(lvl_aligned16(dy) && lvl_aligned16(x) && lvl_aligned16(dw) && lvl_aligned16(ws),
              "linear_wgrad: pointers must be 16-byte aligned");
  const Plan p = make_plan(N, K, M);
  if (!p.ok) return lvl_fail(LVL_ENOSYS, "linear_wgrad: no tiling for N=%d K=%d (multiples of 192/288/384 or 128/256 needed)", N, K);
  hipStream_t st = (hipStream_t)stream;
  float* part = ws;
  float* bpart = dbias ? ws + (size_t)p.S * N * K : nullptr;
  int rc = LVL_OK;
  switch (p.cfg) {
    case 0: rc = launch<4, 2, 6, 6>(p, dy, x, part, bpart, M, N, K, sched, st); break;
    case 1: rc = launch<2, 4, 6, 6>(p, dy, x, part, bpart, M, N, K, sched, st); break;
    case 2: rc = launch<3, 2, 6, 6>(p, dy, x, part, bpart, M, N, K, sched, st); break;
    case 3: rc = launch<2, 3, 6, 6>(p, dy, x, part, bpart, M, N, K, sched, st); break;
    case 4: rc = launch<2, 2, 6, 6>(p, dy, x, part, bpart, M, N, K, sched, st); break;
    case 5: rc = launch<2, 4, 8, 4>(p, dy, x, part, bpart, M, N, K, sched, st); break;
    case 6: rc = launch<2, 2, 8, 4>(p, dy, x, part, bpart, M, N, K, sched, st); break;
    default: rc = launch<1, 4, 8, 4>(p, dy, x, part, bpart, M, N, K, sched, st); break;
  }
  if (rc != LVL_OK) return rc;
  const int64_t NK = (int64_t)N * K;
  int64_t nthreads = (NK + 3) / 4;
  if (nthreads < N) nthreads = N;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, part, bpart,
                     (const uint16_t*)dy, (const uint16_t*)x, dw, dbias, (M / MS) * MS, M, N, K, p.S);
  LVL_CHECK_LAUNCH("linear_wgrad_reduce");
  return LVL_OK;
}
