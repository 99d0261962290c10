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
//   3  y = acc (+ bias) + aux_in                                           proj / fc2 + residual: the block's
//                                                                          `x + attn(...)`, `x + mlp(...)` adds
//                                                                          (timesformer.py:183-196) leave the GEMM as
//                                                                          the new residual stream
//   4  u = acc + bias ; y = u * sigmoid(1.702 u) ; aux_out = quickgelu'(u) epilogue 1 for training (bf16 only): the
//                                                                          backward needs the DERIVATIVE, and the forward
//                                                                          has sigmoid(1.702 u) in a register already
//   5  y = acc * aux_in ; column sums of y -> partial slab                 epilogue 2 on the stored derivative: no
//                                                                          transcendental in the backward (epilogue 2
//                                                                          is VALU-bound: +6 us per 28-us tile)
//
// f32-CLASS MODE (template F32O, C-ABI dtype LVL_F32): the SAME kernel -- tile walk, LDS-DMA ring, swizzle, MFMA phases,
// bias image, epilogue arithmetic -- with float32 results. The operands are bf16 TERM IMAGES of float32 matrices
// (lvl_split_bf16x3, elementwise.hip): x = xh + xl (+ O(2^-18 x)), xh = bf16(x), xl = bf16(x - xh), laid out along the
// contraction as X3 = [xh | xh | xl], W3 = [wh | wl | wh], so that one pass over K' = 3K accumulates
// xh.wh + xh.wl + xl.wh in the f32 accumulators: every product is exact, what is dropped is xl.wl and the second-order
// remainders (~2^-17 relative per product). y / aux_out / aux_in are float32; the QuickGELU epilogues see the unrounded
// f32 pre-activation. This is the parity configuration's GEMM (north_star: "within 1e-3 fp32"): it puts the benched
// kernel itself, not a library GEMM, under the f32 tolerance.
//
// Shape of the problem on this path: M = B*T ~ 2e5 rows, N,K in {768, 2304, 3072}. Measured facts that shaped it
// (profiles/r02_pmc_gemm_tn_v1_qkv.txt, tools/probe_gemm_trace.py): (1) what the L2 can serve is a number of
// REQUESTS, so every request must be a full 128-byte line (a K step of 32 = 64-byte rows doubled the requests and
// ran at 650-930 TF/s); (2) a workgroup that ends after one tile pays ~17 us of launch + first-fetch + store drain
// per 20-us tile; (3) under this load the chip clocks at 1.3-1.7 GHz (power), so instruction and LDS economy matter
// as much as stalls. Design:
//   * PERSISTENT workgroups (one per CU) of 8 waves (2 along M x 4 along N) walk 256x256 output tiles; the K loop
//     never drains at a tile boundary: the first blocks of the next tile are in flight, and its first operands
//     already in registers, while the finished tile is stored;
//   * v_mfma_f32_32x32x16_bf16 with SWAPPED operands (A = weight rows, B = activation rows): the accumulator then
//     holds, per lane, 4 consecutive output columns of one row, so bias / activation / aux tensors are read and
//     written as 8- and 16-byte vectors without an LDS transpose of the tile;
//   * K is walked in blocks of 64 (one 128-byte line per operand row). The 128 KiB of LDS are a ring of 8 SLOTS of
//     128 rows x 128 B: a K block is four slots (X rows 0-127 | X rows 128-255 | W half 0 | W half 1), two blocks are
//     resident. LDS is only a transit buffer: a wave's share of a slot is read ONCE into registers (X: 64 rows = 8
//     fragments, W: 32 rows = 4 fragments) and the slot is refilled by LDS-DMA (global_load_lds_dwordx4, no VGPR round
//     trip) as soon as every wave has consumed it -- 5-6 phases before it is read again;
//   * a K block is four PHASES, one output quadrant (64 x 32 per wave, 8 MFMAs) each, ordered so that every phase
//     needs exactly one new operand: P0 X0.W0, P1 X0.W1, P2 X1.W1, P3 X1.W0. All waves run in step with ONE bare
//     s_barrier per phase; the reads of the NEXT phase's operand and the refill DMAs ride between the MFMAs (one
//     per MFMA), so no LDS or DMA latency separates two phases and the two waves of a SIMD cover each other;
//   * the eight 16-byte chunks of a row are XOR-permuted inside a slot (chunk ^ ((row>>1) & 7), applied to the per-lane
//     SOURCE address of the DMA and to the fragment reads): every ds_read_b128 lane group touches 16 distinct 16-byte
//     slots of the 256-byte bank row (SQ_LDS_BANK_CONFLICT = 0), and the 8 lanes of a row still fetch one whole line;
//   * the tile's bias values travel by LDS-DMA too, so the epilogue issues no vector-memory load; its stores stay in
//     flight across the next tile's first blocks (the vmcnt allowances account for them);
//   * blockIdx -> tile is XCD-aware (workgroup i runs on XCD i % 8): each XCD owns a contiguous range of the
//     N-fastest tile order, so the N/256 tiles that read the same 256 rows of X sit behind one L2 and X comes from
//     HBM once; W (<= 4.7 MB) lives in L2 / Infinity Cache;
//   * inside an XCD's range the tiles are handed out by a DEVICE TILE COUNTER (one word per XCD, `sched`): a
//     workgroup takes the next tile of its XCD's queue with one returning atomic add, issued in front of the
//     result stores of the tile before the tile before, so that its latency never shows (the reply is parked in a register, published to the other waves
//     through an LDS mailbox two K blocks later -- by then the counted vmcnt waits of the K loop have covered it --
//     and read by the fetch cursor when it wraps to the next tile, three K blocks before the end).
//     A compute unit that another kernel holds (an RCCL channel during the gradient all-reduce) therefore costs
//     the launch one tile's worth of throughput, not a static range of tiles: its workgroup starts late, finds
//     the queues drained and leaves. The last workgroup out zeroes the counters again.
#include <type_traits>

#include "common.h"

int lvl_debug_late_mod();
int lvl_colsum_mid_rows();
int lvl_launch_column_reduce(const float* part, int nparts, int width, int seg, float* mid, float* out0, float* out1,
                             float* out2, hipStream_t st);

// the LDS-DMA fills set M0 inside inline asm and say so in the clobber list; this kernel has no other M0 user
#pragma clang diagnostic ignored "-Winline-asm"

typedef __attribute__((ext_vector_type(8))) __bf16 gm_bf16x8;
typedef __attribute__((ext_vector_type(16))) float gm_f32x16;

namespace {

constexpr int BK = 64;                 // contraction elements per K block (one 128-byte line per row)
constexpr int TM = 256, TN = 256;      // workgroup tile
constexpr int SLOT = 128 * 128;        // bytes of one slot: 128 rows x 128 B
constexpr int NSLOT = 8;               // two K blocks x {X0, X1, W0, W1}
constexpr int BIAS_OFF = NSLOT * SLOT; // four 1-KiB images of bias[n0 .. n0+255] behind the ring (tile index & 3)
constexpr int MBOX_OFF = BIAS_OFF + 4 * 1024;   // one word: the tile-queue reply wave 0 publishes to the workgroup
constexpr int SMEM_B = MBOX_OFF + 64;
constexpr int DYN_MIN_NB = 5;          // K blocks a tile needs for the publish (block 2) -> read (block nb-3) hand-off
enum { S_X0 = 0, S_X1 = 1, S_W0 = 2, S_W1 = 3 };

__device__ __forceinline__ gm_f32x16 mfma32(uint4 a, uint4 b, gm_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gm_bf16x8, a), __builtin_bit_cast(gm_bf16x8, b),
                                                 c, 0, 0, 0);
}

// sigmoid through v_exp_f32 + v_rcp_f32 (1 ulp each; the results are rounded to bf16): an IEEE division here costs
// ~10 VALU instructions per element of a 256x256 tile's epilogue
__device__ __forceinline__ float sigmoid1702(float u) { return __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * u)); }
__device__ __forceinline__ float quick_gelu(float u) { return u * sigmoid1702(u); }
__device__ __forceinline__ float quick_gelu_grad(float u) {
  const float s = sigmoid1702(u);
  return s * (1.f + 1.702f * u * (1.f - s));
}

// QuickGELU pieces of the bf16 epilogues, scalar f32 on purpose: inside this dependent chain (cvt -> mul -> exp2 -> add -> rcp ->
// mul, two waves per SIMD) the packed forms v_pk_mul / v_pk_add / v_pk_fma_f32 are SLOWER than the scalar ones (3.87 us scalar,
// 5.03 us packed for the same arithmetic: tools/probes/valu_gelu.hip, profiles/r06_gemm_epilogues.txt; their throughput with
// independent operands is fine, profiles/r06_valu_rates.txt) -- the file is compiled with -fno-slp-vectorize so that the
// compiler does not form them either.
// quickgelu(u) = u r, r = sigmoid(1.702 u) = 1 / (1 + 2^(-1.702 log2(e) u)); quickgelu'(u) = r (1 + 1.702 u (1 - r))
// = r + 1.702 (u r)(1 - r).
__device__ __forceinline__ float bf16_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void qgelu1(float u, float& y, float& r) {
  r = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u * (-1.702f * 1.44269504088896341f)));
  y = u * r;
}
__device__ __forceinline__ float qgelu_grad1(float y, float r) { return fmaf((1.f - r) * y, 1.702f, r); }

// lanes l < 32 and l + 32 hold adjacent 8-byte pieces (4 bf16) of the same output row for two neighbouring
// column groups `a` (columns c..c+3 | c+4..c+7) and `b` (c+8.. | c+12..): one half-swap per dword leaves the
// lower lane with 16 contiguous bytes of group a and the upper lane with 16 contiguous bytes of group b.
__device__ __forceinline__ uint4 widen_pair(uint2 a, uint2 b) {
  const auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
  return make_uint4(r0[0], r1[0], r0[1], r1[1]);
}

// Result stores of the epilogue (bf16 path). GM_STORE_POLICY picks the cache policy of the 16-byte stores (build-time A/B,
// tools/probe_gemm_store_policy.py): 0 plain, 1 nt, 2 sc1, 3 sc0 sc1. What the flavours do on gfx950 (MI355X_MICROARCH.md,
// "stores of each flavour"): plain / nt keep the written line in the XCD's L2, sc1 / sc0 sc1 drop it -- a 256 x 256 tile
// leaves 128-256 KB per workgroup, 4-8 MB per round per XCD, against a 4 MiB L2 that also has to hold the X and W panels.
#ifndef GM_STORE_POLICY
#define GM_STORE_POLICY 0
#endif
__device__ __forceinline__ void gm_store16(void* p, uint4 v) {
#if GM_STORE_POLICY == 0
  *reinterpret_cast<uint4*>(p) = v;
#elif GM_STORE_POLICY == 1
  __builtin_nontemporal_store(__builtin_bit_cast(lvl_u32x4, v), reinterpret_cast<lvl_u32x4*>(p));
#elif GM_STORE_POLICY == 2
  // s_nop 1 inside the string: a store of more than 64 bits reads its data registers for two more wait states, and the
  // hazard recogniser does not look into asm (without it the next VALU write to v corrupts the stored value: measured)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(lvl_u32x4, v)) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(lvl_u32x4, v)) : "memory");
#endif
}

// GM_EXP (timing experiments only, results are WRONG; tools/probe_gemm_variants.py): 1 = no s_barrier in the K loop, 2 = no vmcnt
// wait in front of the barriers, 4 = epilogues 1 / 4 skip their second result store, 8 = epilogue 4 skips the QuickGELU arithmetic
#ifndef GM_EXP
#define GM_EXP 0
#endif

// build-time shape of the aux_in epilogues (see `epilogue`): row groups requested up front, re-read of xA / wA
#ifndef GM_UPFRONT_RES
#define GM_UPFRONT_RES 4
#endif
#ifndef GM_UPFRONT_COLSUM
#define GM_UPFRONT_COLSUM 3
#endif
#ifndef GM_REREAD_RES
#define GM_REREAD_RES 1
#endif

template <int EPI, bool F32O>
__global__ __launch_bounds__(512) void gemm_tn_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                      const float* __restrict__ bias, void* __restrict__ Yv,
                                                      void* __restrict__ aux_out_v,
                                                      const void* __restrict__ aux_in_v,
                                                      float* __restrict__ colpart, int64_t M, int N, int K,
                                                      int tiles_n, int ntiles, unsigned* __restrict__ sched,
                                                      int late_mod) {
  using out_t = std::conditional_t<F32O, float, uint16_t>;      // element type of y / aux_out / aux_in
  out_t* __restrict__ const Y = static_cast<out_t*>(Yv);
  out_t* __restrict__ const aux_out = static_cast<out_t*>(aux_out_v);
  const out_t* __restrict__ const aux_in = static_cast<const out_t*>(aux_in_v);
  extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];      // [NSLOT][128 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nb = K / BK;
  // PERSISTENT workgroups, one per CU: workgroup b sits on XCD b % 8 and works on that XCD's contiguous range
  // [xstart, xstart + cnt) of the N-fastest tile order (bijective for any tile count): at any time the workgroups of
  // an XCD work on neighbouring tiles, so the N/256 tiles that read the same rows of X share one L2.
  //   static schedule (sched == nullptr): tiles xstart + bid/8, + wpx, + 2 wpx, ...
  //   dynamic schedule: tiles xstart + (value of the XCD's counter when the workgroup asked), see the file header.
  const int nx = gridDim.x >= 8 ? 8 : 1;
  const int bid = blockIdx.x, xcd = bid % nx, wpx = gridDim.x / nx;
  const int tq = ntiles / nx, tr = ntiles % nx;
  const int cnt = tq + (xcd < tr ? 1 : 0);
  const int xstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const bool dyn = sched != nullptr;            // the host passes a counter block only when nb >= DYN_MIN_NB
  if (dyn) __builtin_assume(nb >= DYN_MIN_NB);
  if (!dyn && bid / nx >= cnt) return;
  int my_tiles = dyn ? 0x7fffffff : (cnt - bid / nx + wpx - 1) / wpx;
  int pair_even = xstart + bid / nx, pair_odd = pair_even;    // tile index (N-fastest order) of tile ordinal i, by i & 1
  auto set_pair = [&](int i, int v) { if (i & 1) pair_odd = v; else pair_even = v; };
  auto get_pair = [&](int i) { return (i & 1) ? pair_odd : pair_even; };
  // wave 0 / lane 0 of a dynamic workgroup: the counter reply in flight. Written by the memory system when the atomic
  // returns, NOT at the asm statement that issues it: it is only ever read by `publish` below, behind a vmcnt wait
  // that covers the atomic (tests/test_boundary_cpu.py checks in the ISA that nothing else touches the register).
  uint32_t pend = 0;
  unsigned* const ctr = sched + xcd;
  // (lane 0 only: EXEC is narrowed inside the asm -- a divergent C++ branch here would make the compiler treat the
  // loop-carried tile cursor as divergent and move the DMA base addresses into vector registers)
  uint64_t exec_save;
  auto pull = [&]() {
    if (wave == 0)
      asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                   : "=v"(pend), "=&s"(exec_save) : "v"(0u), "v"(1u), "s"(ctr) : "memory");
  };
  auto publish = [&]() {
    if (wave == 0)
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0"
                   : "=&s"(exec_save) : "v"((uint32_t)(uintptr_t)smem + MBOX_OFF), "v"(pend) : "memory");
  };
  // (an LDS read in asm: through a volatile C++ pointer the compiler emits a FLAT load and waits vmcnt(0) for it -- a
  // full drain of the run-ahead fills once per tile, 2-4 % of a launch on the dynamic schedule until round 6)
  auto mailbox = [&]() -> int {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)smem + MBOX_OFF) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
  };
  const bool late = dyn && late_mod > 0 && bid % late_mod == 1;      // test hook (lvl_debug_late_workgroups)
  // first tile: the one synchronous hand-out of a launch. Requested here, consumed behind the address set-up below
  // (nothing else is in flight yet, so the round trip overlaps ~1 us of scalar / vector arithmetic).
  if (dyn && !late) pull();

  // Tile order: all N/256 column tiles of a row block are neighbours (N-fastest). (Panels of 3-6 column tiles, to keep
  // a weight panel L2-resident, measured no faster: the weight re-reads are served by the Infinity Cache.)
  auto decode_tile = [&](int pair, int& tm, int& tn) {
    tm = pair / tiles_n;
    tn = pair - tm * tiles_n;
  };

  // ---- LDS-DMA plan -----------------------------------------------------------------------------------------------
  // A slot is 16 fills of 1 KiB (8 rows x 128 B); wave w issues fills w and w+8 of every slot. Lane l of fill f
  // lands at slot byte f*1024 + l*16 = slot row 8f + (l>>3), physical chunk l&7, and therefore fetches the LOGICAL
  // chunk (l&7) ^ ((row>>1)&7) of its row. Slot rows: X slot q = tile rows q*128 + r; W slot q = tile columns
  // (r>>5)*64 + q*32 + (r&31) (the two 32-column groups of every wave's 64 output columns go to different halves).
  // Per-lane sources are 32-bit byte offsets relative to the tile; the tile and the K block ride in scalars.
  uint32_t xrel[4], wrel[2], xclamp;
  {
    const int sub = lane >> 3;
    const int chunk = (lane & 7) ^ ((4 * wave + (sub >> 1)) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int lr = 8 * (wave + 8 * h) + sub;
#pragma unroll
      for (int q = 0; q < 2; ++q) xrel[q * 2 + h] = (uint32_t)(q * 128 + lr) * (uint32_t)(K * 2) + chunk * 16;
      wrel[h] = (uint32_t)((lr >> 5) * 64 + (lr & 31)) * (uint32_t)(K * 2) + chunk * 16;
    }
    xclamp = (uint32_t)((M - 1) * K * 2) + chunk * 16;       // tail tile: re-read the last row (never stored)
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // fetch cursor: the K block the next refills bring in (two blocks ahead of the one being multiplied)
  int f_i = 0, f_j = 0;
  uint32_t f_xrow;                 // byte offset of the fetch tile's first X row
  const char* f_wbase;             // first W row of the fetch tile
  auto fetch_tile = [&](int i) {
    if (!dyn) set_pair(i, xstart + bid / nx + i * wpx);
    const int pair = __builtin_amdgcn_readfirstlane(get_pair(i));
    int tm, tn;
    decode_tile(pair, tm, tn);
    f_xrow = (uint32_t)tm * (uint32_t)(TM * K * 2);
    f_wbase = reinterpret_cast<const char*>(W) + (int64_t)tn * TN * K * 2;
    if (EPI != 2 && bias != nullptr && wave == 0) {
      // the tile's 256 bias values travel the same way (one 1-KiB LDS-DMA by wave 0, >= 8 phases before the
      // epilogue that reads them): the epilogue then issues no vector-memory LOAD, so nothing in it has to wait
      // for the run-ahead fills or for its own stores
      const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + BIAS_OFF + (uint32_t)(i & 3) * 1024);
      const uint32_t off = (uint32_t)lane * 16;
      const char* base = reinterpret_cast<const char*>(bias + tn * TN);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(off), "s"(base)
                   : "memory", "m0");
    }
  };
  auto fetch_advance = [&]() {
    if (++f_j == nb) {
      if (dyn && f_i + 1 < my_tiles) {
        // the reply wave 0 published at K block 2 of the tile being multiplied (>= 4 barriers ago)
        const int v = mailbox();
        if (v < cnt) {
          set_pair(f_i + 1, xstart + v);
        } else {
          my_tiles = f_i + 1;      // the XCD's queue is drained
        }
      }
      if (f_i + 1 < my_tiles) {
        f_j = 0;
        fetch_tile(++f_i);
      } else {
        f_j = nb - 1;              // past the end: keep re-reading the last block (never consumed)
      }
    }
  };
  // Fills go through inline asm: the compiler's LDS-DMA alias tracking would otherwise put s_waitcnt vmcnt(0) in
  // front of every LDS read and drain the run-ahead.
  auto issue_fill = [&](int type, int par, int h) {
    const char* base = type < S_W0 ? reinterpret_cast<const char*>(X) + (int64_t)f_j * (BK * 2)
                                   : f_wbase + (int64_t)f_j * (BK * 2) + (type == S_W1 ? (int64_t)32 * K * 2 : 0);
    uint32_t off;
    if (type < S_W0) {
      off = xrel[type * 2 + h] + f_xrow;
      off = off < xclamp ? off : xclamp;
    } else {
      off = wrel[h];
    }
    const uint32_t m0v =
        __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(par + type) * SLOT + (uint32_t)wave * 1024 + h * 8192);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(off), "s"(base)
                 : "memory", "m0");
  };
  auto issue_group = [&](int type, int par) {
    issue_fill(type, par, 0);
    issue_fill(type, par, 1);
  };

  // ---- fragment addresses -----------------------------------------------------------------------------------------
  // 32x32x16 operand: lane l carries row (l & 31), contraction elements 8*(l>>5) .. +7 of the K=16 slice kk, i.e.
  // logical chunk 2*kk + (l>>5) of the row's eight 16-byte chunks.
  const int r5 = lane & 31, hi = lane >> 5;
  const int r5_ = r5, hi_ = hi;
  uint32_t xa[4], wa[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int offk = ((2 * kk + hi) ^ ((r5 >> 1) & 7)) * 16;
    xa[kk] = (uint32_t)((wm * 64 + r5) * 128 + offk);
    wa[kk] = (uint32_t)((wn * 32 + r5) * 128 + offk);
  }
  auto read_x = [&](int slot, uint4 (&xf)[8]) {
    const uint8_t* b = smem + slot * SLOT;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) xf[mt * 4 + kk] = *reinterpret_cast<const uint4*>(b + xa[kk] + mt * 4096);
  };
  auto read_w = [&](int slot, uint4 (&wf)[4]) {
    const uint8_t* b = smem + slot * SLOT;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wf[kk] = *reinterpret_cast<const uint4*>(b + wa[kk]);
  };

  gm_f32x16 acc[2][2][2];           // [qm][qn][mt]
  // The accumulators of tile ti START as the tile's bias (read straight from its LDS image into the accumulator
  // registers: no VALU work), so the epilogue has no bias add; without a bias they start at zero.
  auto init_acc = [&](int ti) {
    if (EPI != 2 && bias != nullptr) {
      const uint8_t* bias_img = smem + BIAS_OFF + (ti & 3) * 1024;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4 b = *reinterpret_cast<const float4*>(bias_img + (wn * 64 + i * 32 + 8 * rq + 4 * hi) * 4);
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            acc[a >> 1][i][a & 1][4 * rq + 0] = b.x;
            acc[a >> 1][i][a & 1][4 * rq + 1] = b.y;
            acc[a >> 1][i][a & 1][4 * rq + 2] = b.z;
            acc[a >> 1][i][a & 1][4 * rq + 3] = b.w;
          }
        }
    } else {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a >> 2][(a >> 1) & 1][a & 1][r] = 0.f;
    }
  };

  uint4 xA[8], xB[8], wA[4], wB[4];      // operand fragments of the K loop (declared here: the epilogue re-reads xA / wA)
  int par = 0;                            // ring half (slot offset 0 / 4) of the K block being multiplied

  auto epilogue = [&](int tm, int tn, int ti) {
    const int64_t m0 = (int64_t)tm * TM;
    const int n0 = tn * TN;
    // the lane's row / column-half as the epilogue sees them: opaque copies, so that the per-lane 64-bit row offsets
    // (tile-invariant: (wm*64 + r5) * N + ...) are computed here, per tile, instead of being hoisted out of the tile
    // loop into registers the K loop does not have (they were spilled, and reloaded BEHIND the result stores)
    int r5 = r5_, hi = hi_;
    asm volatile("" : "+v"(r5), "+v"(hi));
    constexpr bool COLSUM = EPI == 2 || EPI == 5;                 // column sums of y leave as a partial slab
    constexpr bool AUXIN = EPI == 2 || EPI == 3 || EPI == 5;      // an aux_in row rides with every result row
    constexpr bool TWO_OUT = EPI == 1 || EPI == 4;
    float csum[COLSUM ? 32 : 1];          // value c = column (c>>4)*32 + ((c>>2)&3)*8 + 4*hi + (c&3) of the wave's 64
    if (COLSUM) {
#pragma unroll
      for (int c = 0; c < 32; ++c) csum[c] = 0.f;
    }
    if constexpr (F32O) {
      // float32 results (f32-class mode): an accumulator quad IS 4 consecutive output columns of one row -> one
      // 16-byte store per quad (and one 16-byte aux access for the QuickGELU epilogues); same arithmetic as below
      // without the bf16 roundings
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t m = m0 + (j >> 1) * 128 + wm * 64 + (j & 1) * 32 + r5;
        const bool valid = m < M;
        const int64_t rowoff = (valid ? m : M - 1) * (int64_t)N + n0 + wn * 64 + 4 * hi;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const gm_f32x16& a16 = acc[j >> 1][i][j & 1];
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int64_t o = rowoff + i * 32 + 8 * rq;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a16[4 * rq + e];
            if (EPI == 1) {
              if (valid) *reinterpret_cast<float4*>(aux_out + o) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
            }
            if (EPI == 3) {
              const float4 r = *reinterpret_cast<const float4*>(aux_in + o);
              v[0] += r.x;
              v[1] += r.y;
              v[2] += r.z;
              v[3] += r.w;
            }
            if (EPI == 2) {
              const float4 u = *reinterpret_cast<const float4*>(aux_in + o);
              v[0] *= quick_gelu_grad(u.x);
              v[1] *= quick_gelu_grad(u.y);
              v[2] *= quick_gelu_grad(u.z);
              v[3] *= quick_gelu_grad(u.w);
              if (valid) {
#pragma unroll
                for (int e = 0; e < 4; ++e) csum[i * 16 + rq * 4 + e] += v[e];
              }
            }
            if (valid) *reinterpret_cast<float4*>(Y + o) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    } else {
    // bf16 results. The hardware's vmcnt counts result stores; the compiler's wait insertion does not count stores that sit
    // under a branch (the `m < M` guard: its counters are merged conservatively across it): a wait it
    // places for an aux_in load (residual stream / QuickGELU derivative / pre-activation rows) BEHIND a group's stores is
    // too strict by the number of those stores and waits for their acknowledgement -- one store drain per row group, +10 us
    // per tile with the loads one group ahead of the stores (round 6: proj + residual 0.34 ms against 0.23 ms for the plain
    // epilogue on the same shape). So the aux rows of ALL FOUR row groups are requested, and every one of them has landed
    // (`landed`: a register use the compiler has to wait for), before the first store leaves; from there on nothing is
    // pending and packing and stores interleave freely. Registers: UPFRONT groups are requested at once, the rest when
    // group 0 is packed and its accumulators are dead; HOLD groups are packed before the first store; with REREAD the next
    // tile's first fragments (xA / wA, read during the last K block) are not carried across the epilogue but read again
    // from the ring under the last stores (the column-sum epilogues hold 32 running sums on top of the aux rows).
    constexpr int UPFRONT = COLSUM ? GM_UPFRONT_COLSUM : GM_UPFRONT_RES;
    constexpr int HOLD = UPFRONT == 4 ? 1 : 2;
    constexpr bool REREAD = COLSUM ? true : (GM_REREAD_RES != 0);
    uint2 ub[AUXIN ? 4 : 1][8];
    auto load_u = [&](int j, uint2 (&dst)[8]) {
      const int64_t m = m0 + (j >> 1) * 128 + wm * 64 + (j & 1) * 32 + r5;
      const uint16_t* urow = reinterpret_cast<const uint16_t*>(aux_in) + (m < M ? m : M - 1) * (int64_t)N + n0 + wn * 64 + 4 * hi;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) dst[i * 4 + rq] = *reinterpret_cast<const uint2*>(urow + i * 32 + 8 * rq);
    };
    auto landed = [&](const uint2 (&src)[8]) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" ::"v"(src[q].x), "v"(src[q].y));
    };
    if constexpr (AUXIN) {
#pragma unroll
      for (int j = 0; j < UPFRONT; ++j) load_u(j, ub[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    uint4 yv[AUXIN ? HOLD : 1][2][2], uv[2][2];      // [row group][qn][jj]: 16 bytes = columns qn*32 + 16*jj + 8*hi .. +7 of row r5
    auto store_group = [&](int j, const uint4 (&y4)[2][2]) {
      // 16-byte stores: lower lanes take columns 16*jj .. +7, upper lanes 16*jj + 8 .. +15 of a column group
      // (an LDS-staged variant that leaves as full 128-byte lines measured no faster: the tail is not line-bound)
      const int64_t m = m0 + (j >> 1) * 128 + wm * 64 + (j & 1) * 32 + r5;
      if (m < M) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int64_t o = m * (int64_t)N + n0 + wn * 64 + i * 32 + 16 * jj + 8 * hi;
            gm_store16(Y + o, y4[i][jj]);
            if (TWO_OUT && !(GM_EXP & 4)) gm_store16(aux_out + o, uv[i][jj]);      // (GM_EXP 4: timing experiment)
          }
      }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // row group (qm, mt): 32 rows
      const int64_t mg = m0 + (j >> 1) * 128 + wm * 64 + (j & 1) * 32;
      const float rowv = mg + r5 < M ? 1.f : 0.f;      // rows behind M (tail tile) stay out of the column sums
      uint4 (&y4)[2][2] = yv[AUXIN && j < HOLD ? j : 0];
#pragma unroll
      for (int i = 0; i < 2; ++i) {        // column group qn
        const gm_f32x16& a16 = acc[j >> 1][i][j & 1];
        uint2 ypk[4], upk[4];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = a16[4 * rq + e];
          if (EPI == 1) {
            upk[rq] = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
            // the activation sees the ROUNDED pre-activation (what the reference's bf16 Linear output holds and
            // what the backward reads back)
            float r;
            qgelu1(bf16_lo(upk[rq].x), v[0], r);
            qgelu1(bf16_hi(upk[rq].x), v[1], r);
            qgelu1(bf16_lo(upk[rq].y), v[2], r);
            qgelu1(bf16_hi(upk[rq].y), v[3], r);
          }
          if (EPI == 4) {                // y = quickgelu(u), aux_out = quickgelu'(u): one exp2 + one rcp serve both
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float y, r;
              if (GM_EXP & 8) { y = v[e]; r = v[e]; } else qgelu1(v[e], y, r);      // (GM_EXP 8: timing experiment)
              g[e] = (GM_EXP & 8) ? r : qgelu_grad1(y, r);
              v[e] = y;
            }
            upk[rq] = make_uint2(f32x2_to_bf16x2(g[0], g[1]), f32x2_to_bf16x2(g[2], g[3]));
          }
          if (AUXIN) {
            const uint2 ab = ub[j][i * 4 + rq];
            const float a4[4] = {bf16_lo(ab.x), bf16_hi(ab.x), bf16_lo(ab.y), bf16_hi(ab.y)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (EPI == 3) v[e] += a4[e];        // + the residual row (bf16), in f32 before the one rounding of the sum
              if (EPI == 5) v[e] *= a4[e];        // the forward left quickgelu'(u) itself
              if (EPI == 2) {
                float y, r;
                qgelu1(a4[e], y, r);
                v[e] *= qgelu_grad1(y, r);
              }
            }
          }
          if (COLSUM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[i * 16 + rq * 4 + e] = fmaf(v[e], rowv, csum[i * 16 + rq * 4 + e]);
          }
          ypk[rq] = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          y4[i][jj] = widen_pair(ypk[2 * jj], ypk[2 * jj + 1]);
          if (TWO_OUT) uv[i][jj] = widen_pair(upk[2 * jj], upk[2 * jj + 1]);
        }
      }
      if constexpr (AUXIN) {
        if (j == 0 && UPFRONT < 4) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jn = UPFRONT; jn < 4; ++jn) load_u(jn, ub[jn]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j == HOLD - 1) {               // every aux row has landed: stores may leave
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jn = HOLD; jn < 4; ++jn) landed(ub[jn]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jp = 0; jp < HOLD; ++jp) store_group(jp, yv[jp]);
        }
        if (REREAD && j == 3) {
          __builtin_amdgcn_sched_barrier(0);
          read_x(par + S_X0, xA);
          read_w(par + S_W0, wA);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j >= HOLD) store_group(j, y4);
      } else {
        store_group(j, y4);
      }
    }
    }
    if (COLSUM) {
      // column sums over the wave's 128 rows: 32 values per lane, summed over the 32 lanes of each half-wave by a
      // halving butterfly (31 exchanges): lane r5 ends up with the total of value index c = r5
      float (&cs)[COLSUM ? 32 : 1] = csum;
      int cnt2 = 16;
#pragma unroll
      for (int mask = 16; mask >= 1; mask >>= 1) {
        const bool up = (lane & mask) != 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (c < cnt2) {
            const float lo = cs[c], hv = cs[c + cnt2];
            const float send = up ? lo : hv, keep = up ? hv : lo;
            cs[c] = keep + __shfl_xor(send, mask, 64);
          }
        }
        cnt2 >>= 1;
      }
      const int c = r5;
      const int col = (c >> 4) * 32 + ((c >> 2) & 3) * 8 + 4 * hi + (c & 3);
      colpart[(size_t)(tm * 2 + wm) * N + n0 + wn * 64 + col] = cs[0];
    }
  };

#ifdef GM_TRACE
  // debug build only (tools/probe_gemm_trace.py): wall-clock stamps (100 MHz) of wave 0 and wave 4 of every workgroup
  // (the column-sum epilogues keep their partial slab in front of the stamps: [2 * tiles_m][N] floats)
  unsigned long long* trace = reinterpret_cast<unsigned long long*>(colpart + ((EPI == 2 || EPI == 5) ? (size_t)2 * (ntiles / tiles_n) * N : 0)) +
                              ((size_t)bid * 2 + wm) * 128;
  int tpos = 0;
#define GM_STAMP()                                                          \
  do {                                                                      \
    if (wn == 0 && lane == 0 && tpos < 64) {                                \
      trace[64 + tpos] = __builtin_readcyclecounter();                      \
      trace[tpos++] = wall_clock64();                                       \
    }                                                                       \
  } while (0)
#else
#define GM_STAMP() do { } while (0)
#endif
  if (late) {
    my_tiles = 0;                               // as if the queues were drained
  } else if (dyn) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int v = mailbox();
    if (v >= cnt) my_tiles = 0;                 // queue already drained (this workgroup got its CU late)
    pair_even = xstart + v;
    __builtin_amdgcn_s_barrier();               // everyone has read the mailbox before it is written again
    if (my_tiles) pull();                       // tile 1: the reply lands during the prologue fills and tile 0's first blocks
  }
  GM_STAMP();
  if (my_tiles > 0) {
  fetch_tile(0);
#pragma unroll
  for (int b0 = 0; b0 < 2; ++b0) {         // blocks 0 and 1, groups in the order the phases will read them
    issue_group(S_X0, b0 * 4);
    issue_group(S_W0, b0 * 4);
    issue_group(S_W1, b0 * 4);
    issue_group(S_X1, b0 * 4);
    fetch_advance();
  }
  // LOCKSTEP schedule. All 8 waves run the same phase; ONE barrier per phase. A phase = 8 MFMAs per wave with, riding
  // between them, the fragment reads of the operand the NEXT phase needs and this phase's refill DMAs; the two waves
  // of a SIMD interleave freely inside a phase, so one wave's LDS / DMA issue hides under the other's MFMAs:
  //   P0 = X0.W0  reads W1(j)            no refill
  //   P1 = X0.W1  reads X1(j)            refills X0(j), W0(j) <- block j+2   (both consumed in P0)
  //   P2 = X1.W1  reads X0(j+1)          refills W1(j)        <- block j+2   (consumed in P1)
  //   P3 = X1.W0  reads W0(j+1)          refills X1(j)        <- block j+2   (consumed in P2)
  // (W0 fragment k is re-read right after its last use in P3.) A slot is refilled only after a barrier that follows
  // the phase in which every wave CONSUMED its fragments (the compiler's lgkmcnt wait in front of the consuming MFMA
  // is the completion of the read), and is read again 5-6 phases after the refill. Before each barrier a wave waits
  // until the fills of the slot the next phase reads have landed: fills return in order, so "all but the N newest
  // vector-memory operations" with N = the operations issued after that slot's fills (10 / 8 / 10 / 10 for P0..P3;
  // the bias image and the epilogue's stores only make the wait stricter). The barrier is the bare s_barrier: a
  // fence would drain the run-ahead fills.
#define GM_BAR(N)                                                     \
  do {                                                                \
    if (!(GM_EXP & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");          \
    if (!(GM_EXP & 1)) __builtin_amdgcn_s_barrier();                  \
    asm volatile("" ::: "memory");                                    \
    __builtin_amdgcn_sched_barrier(0);                                \
  } while (0)
  // 8 MFMAs; after MFMA k: rd(k) (one fragment read) and dm(k) (one refill DMA or nothing)
#define GM_PHASE(xf, wf, c, rd, dm)                                                                       \
  do {                                                                                                    \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                    \
      _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                                  \
        c[mt] = mfma32(wf[kk], xf[mt * 4 + kk], c[mt]);                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        rd(kk * 2 + mt);                                                                                  \
        dm(kk * 2 + mt);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
  GM_BAR(12);                            // X0 and W0 of block 0 have landed
  read_x(S_X0, xA);
  read_w(S_W0, wA);
  __builtin_amdgcn_sched_barrier(0);
  GM_STAMP();
  init_acc(0);
  // One K block = four phases. A0 / A1 are added to the vmcnt allowance of the barriers in front of P0,P1 / P2,P3:
  // vector-memory operations retire in issue order, so in the first block(s) after an epilogue -- while the slot a
  // barrier waits for was still filled BEFORE that epilogue -- the epilogue's NS result stores sit between the
  // awaited fills and the newer ones and may stay in flight (waiting for them to be acknowledged would cost every
  // tile ~2 us of idle matrix pipe).
  // (f32-class mode: its 32 / 64 wider stores would overflow the 6-bit vmcnt field together with the fills -- no
  // allowance there: the first blocks of the next tile also wait for the previous tile's stores)
  constexpr int NS = F32O ? 0 : (EPI == 1 || EPI == 4 ? 32 : 16);
  auto k_block = [&](auto a0_, auto a1_) {
    constexpr int A0 = decltype(a0_)::value, A1 = decltype(a1_)::value;
    const uint8_t* sb = smem + par * SLOT;
    const uint8_t* sbn = smem + (par ^ 4) * SLOT;
    auto rd_w1 = [&](int k) { if (k < 4) wB[k] = *reinterpret_cast<const uint4*>(sb + S_W1 * SLOT + wa[k]); };
    auto rd_x1 = [&](int k) { xB[k] = *reinterpret_cast<const uint4*>(sb + S_X1 * SLOT + xa[k & 3] + (k >> 2) * 4096); };
    auto rd_x0 = [&](int k) { xA[k] = *reinterpret_cast<const uint4*>(sbn + S_X0 * SLOT + xa[k & 3] + (k >> 2) * 4096); };
    // W0 of the next block: fragment kk is free once both MFMAs of step kk have been issued (k = 2*kk + 1)
    auto rd_w0 = [&](int k) { if (k & 1) wA[k >> 1] = *reinterpret_cast<const uint4*>(sbn + S_W0 * SLOT + wa[k >> 1]); };
    auto dm_none = [&](int) {};
    auto dm_x0w0 = [&](int k) {
      if (k == 0) issue_fill(S_X0, par, 0);
      if (k == 2) issue_fill(S_X0, par, 1);
      if (k == 4) issue_fill(S_W0, par, 0);
      if (k == 6) issue_fill(S_W0, par, 1);
    };
    auto dm_w1 = [&](int k) {
      if (k == 1) issue_fill(S_W1, par, 0);
      if (k == 5) issue_fill(S_W1, par, 1);
    };
    auto dm_x1 = [&](int k) {
      if (k == 0) issue_fill(S_X1, par, 0);
      if (k == 4) issue_fill(S_X1, par, 1);
    };
    GM_BAR(10 + A0);
    GM_PHASE(xA, wA, acc[0][0], rd_w1, dm_none);
    GM_BAR(8 + A0);
    GM_PHASE(xA, wB, acc[0][1], rd_x1, dm_x0w0);
    GM_BAR(10 + A1);
    GM_PHASE(xB, wB, acc[1][1], rd_x0, dm_w1);
    GM_BAR(10 + A1);
    GM_PHASE(xB, wA, acc[1][0], rd_w0, dm_x1);
    fetch_advance();
    par ^= 4;
  };
  using I0 = std::integral_constant<int, 0>;
  using IS = std::integral_constant<int, NS>;
  for (int i = 0; i < my_tiles; ++i) {
    int j = 0;
    if (i > 0) {                     // the slots read here were filled before the previous tile's epilogue
      k_block(IS{}, IS{});
      if (nb > 1) k_block(IS{}, I0{});
      j = 2;
    }
    for (; j < nb; ++j) {
      if (dyn && j == 2) {
        // K block 2: the counter reply asked for before the previous tile's epilogue is an epilogue and 16 fills old;
        // "all but the 10 newest" -- the wait the next barrier performs anyway -- has it in its register
        if (wave == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        publish();
      }
      k_block(I0{}, I0{});
    }
    // the fills of the next tile's first blocks are in flight and its first X0 / W0 fragments already in registers
    // while this tile's results leave
    GM_STAMP();
    // Ask for the tile after the next one NOW, in front of this tile's result stores: vector-memory replies come back
    // in issue order, so a slow device-scope atomic holds up the completion count of every fill issued behind it --
    // here the fills in front of it are the ones the next barriers wait for (the next tile's blocks 0 and 1 were
    // fetched before this point), and the epilogue (~3 us) plus two K blocks pass before anything behind it is awaited.
    // (Issued at the top of a tile the reply delayed that tile's first fills: +3 % on the K = 768 shapes.)
    if (dyn && my_tiles == 0x7fffffff) pull();
    const int pair = get_pair(i);
    int tm, tn;
    decode_tile(pair, tm, tn);
    epilogue(tm, tn, i);
    GM_STAMP();
    init_acc(i + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  }
#undef GM_BAR
#undef GM_PHASE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the run-ahead fills before this workgroup's LDS is freed
  // the last pull's reply is never consumed: keep its register reserved until the drain above has delivered it
  asm volatile("" ::"v"(pend));
  if (dyn && wave == 0 && lane == 0) {
    // every workgroup of the launch signs off once (its counter atomics are complete: vmcnt(0) above); the last one
    // out leaves the counter block zeroed for the next launch that is handed the same block
    const unsigned gone = __hip_atomic_fetch_add(sched + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == gridDim.x - 1) {
#pragma unroll
      for (int q = 0; q < 9; ++q) __hip_atomic_store(sched + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int num_cus() { return lvl_persistent_cus(); }      // one persistent workgroup per compute unit

template <int EPI, bool F32O = false>
int launch_tn(const void* x, const void* w, const float* bias, void* y, void* aux_out, const void* aux_in,
              float* colpart, int64_t M, int N, int K, unsigned* sched, hipStream_t st) {
  constexpr int shmem = SMEM_B;
  const int rc = lvl_allow_lds<gemm_tn_kernel<EPI, F32O>>();
  if (rc != LVL_OK) return rc;
  const int tiles_n = N / TN;
  const int64_t tiles_m = (M + TM - 1) / TM;
  const int64_t ntiles = tiles_m * tiles_n;
  if (ntiles > 0x7fffffff) return lvl_fail(LVL_EINVAL, "linear_tn: too many tiles");
  int grid = (int)(ntiles < num_cus() ? ntiles : num_cus());
  if (grid >= 8) grid -= grid % 8;          // whole XCD rounds (the kernel maps workgroup b to XCD b % 8)
  if (K / BK < DYN_MIN_NB) sched = nullptr; // too few K blocks per tile for the counter hand-off: static schedule
  hipLaunchKernelGGL((gemm_tn_kernel<EPI, F32O>), dim3((unsigned)grid), dim3(512), shmem, st, (const uint16_t*)x,
                     (const uint16_t*)w, bias, y, aux_out, aux_in, colpart, M,
                     N, K, tiles_n, (int)ntiles, sched, sched ? lvl_debug_late_mod() : 0);
  LVL_CHECK_LAUNCH("linear_tn");
  return LVL_OK;
}

}  // namespace

int64_t lvl_linear_tn_workspace_floats(int64_t M, int64_t N) {
  return (2 * ((M + TM - 1) / TM) + lvl_colsum_mid_rows()) * N;      // column partials + the reducer's intermediate rows
}

#ifdef GM_TRACE
extern "C" int lvl_linear_tn_trace(const void* x, const void* w, const float* bias, void* y, void* trace, int64_t M,
                                   int N, int K, void* stream) {
  return launch_tn<0>(x, w, bias, y, nullptr, nullptr, (float*)trace, M, N, K, nullptr, (hipStream_t)stream);
}
// the same with an epilogue (0, 1, 3, 4, 5); epilogue 5: `trace` starts with room for the column partials
extern "C" int lvl_linear_tn_trace_epi(const void* x, const void* w, const float* bias, void* y, void* aux_out,
                                       const void* aux_in, void* trace, int64_t M, int N, int K, int epilogue, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case 0: return launch_tn<0>(x, w, bias, y, nullptr, nullptr, (float*)trace, M, N, K, nullptr, st);
    case 1: return launch_tn<1>(x, w, bias, y, aux_out, nullptr, (float*)trace, M, N, K, nullptr, st);
    case 3: return launch_tn<3>(x, w, bias, y, nullptr, aux_in, (float*)trace, M, N, K, nullptr, st);
    case 4: return launch_tn<4>(x, w, bias, y, aux_out, nullptr, (float*)trace, M, N, K, nullptr, st);
    case 5: return launch_tn<5>(x, w, nullptr, y, nullptr, aux_in, (float*)trace, M, N, K, nullptr, st);
  }
  return -1;
}
#endif

extern "C" int lvl_linear_tn(const void* x, const void* w, const float* bias, void* y, void* aux_out,
                             const void* aux_in, float* colsum, float* ws, uint32_t* sched, int64_t M, int N, int K,
                             int epilogue, int dtype, void* stream) {
  LVL_REQUIRE(x && w && y, "linear_tn: null pointer");
  LVL_REQUIRE(dtype == LVL_BF16 || dtype == LVL_F32, "linear_tn: unknown dtype %d", dtype);
  LVL_REQUIRE(M > 0 && N > 0 && K > 0, "linear_tn: empty problem");
  if (N % TN != 0 || K % BK != 0)
    return lvl_fail(LVL_ENOSYS, "linear_tn: no tiling for N=%d K=%d (N %% 256 == 0 and K %% 64 == 0 needed)", N, K);
  if ((uint64_t)M * K * 2 >= (1ull << 32) || (uint64_t)N * K * 2 >= (1ull << 32))
    return lvl_fail(LVL_ENOSYS, "linear_tn: operand larger than 4 GiB (32-bit DMA offsets)");
  LVL_REQUIRE(lvl_aligned16(x) && lvl_aligned16(w) && lvl_aligned16(y) && lvl_aligned16(bias) &&
                  lvl_aligned16(aux_out) && lvl_aligned16(aux_in),
              "linear_tn: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LVL_F32) {      // f32-class mode: bf16 term images in (K = 3 x the layer's width), float32 out
    LVL_REQUIRE(K % (3 * BK) == 0, "linear_tn (f32 class): K = %d is not 3 x a multiple of 64", K);
    switch (epilogue) {
      case LVL_EPI_BIAS:
        return launch_tn<0, true>(x, w, bias, y, nullptr, nullptr, nullptr, M, N, K, sched, st);
      case LVL_EPI_BIAS_QUICKGELU:
        LVL_REQUIRE(aux_out != nullptr, "linear_tn: the QuickGELU epilogue writes the pre-activation to aux_out");
        return launch_tn<1, true>(x, w, bias, y, aux_out, nullptr, nullptr, M, N, K, sched, st);
      case LVL_EPI_BIAS_RESIDUAL:
        LVL_REQUIRE(aux_in != nullptr, "linear_tn: the residual epilogue reads the residual rows from aux_in");
        return launch_tn<3, true>(x, w, bias, y, nullptr, aux_in, nullptr, M, N, K, sched, st);
      case LVL_EPI_QUICKGELU_BWD: {
        LVL_REQUIRE(aux_in != nullptr && colsum != nullptr && ws != nullptr && lvl_aligned16(ws),
                    "linear_tn: the QuickGELU-backward epilogue needs aux_in, colsum and a workspace");
        const int rc = launch_tn<2, true>(x, w, nullptr, y, nullptr, aux_in, ws, M, N, K, sched, st);
        if (rc != LVL_OK) return rc;
        const int P = (int)(2 * ((M + TM - 1) / TM));
        return lvl_launch_column_reduce(ws, P, N, N, ws + (size_t)P * N, colsum, nullptr, nullptr, st);
      }
      case LVL_EPI_BIAS_QUICKGELU_DERIV:
      case LVL_EPI_MUL_AUX_COLSUM:
        return lvl_fail(LVL_ENOSYS, "linear_tn: epilogue %d is a bf16 epilogue (the f32-class mode keeps the pre-activation: "
                                    "LVL_EPI_BIAS_QUICKGELU / LVL_EPI_QUICKGELU_BWD)", epilogue);
      default:
        return lvl_fail(LVL_EINVAL, "linear_tn: unknown epilogue %d", epilogue);
    }
  }
  switch (epilogue) {
    case LVL_EPI_BIAS:
      return launch_tn<0>(x, w, bias, y, nullptr, nullptr, nullptr, M, N, K, sched, st);
    case LVL_EPI_BIAS_QUICKGELU:
      LVL_REQUIRE(aux_out != nullptr, "linear_tn: the QuickGELU epilogue writes the pre-activation to aux_out");
      return launch_tn<1>(x, w, bias, y, aux_out, nullptr, nullptr, M, N, K, sched, st);
    case LVL_EPI_BIAS_RESIDUAL:
      LVL_REQUIRE(aux_in != nullptr, "linear_tn: the residual epilogue reads the residual rows from aux_in");
      return launch_tn<3>(x, w, bias, y, nullptr, aux_in, nullptr, M, N, K, sched, st);
    case LVL_EPI_QUICKGELU_BWD: {
      LVL_REQUIRE(aux_in != nullptr && colsum != nullptr && ws != nullptr && lvl_aligned16(ws),
                  "linear_tn: the QuickGELU-backward epilogue needs aux_in, colsum and a workspace");
      const int rc = launch_tn<2>(x, w, nullptr, y, nullptr, aux_in, ws, M, N, K, sched, st);
      if (rc != LVL_OK) return rc;
      const int P = (int)(2 * ((M + TM - 1) / TM));
      return lvl_launch_column_reduce(ws, P, N, N, ws + (size_t)P * N, colsum, nullptr, nullptr, st);
    }
    case LVL_EPI_BIAS_QUICKGELU_DERIV:
      LVL_REQUIRE(aux_out != nullptr, "linear_tn: the QuickGELU + derivative epilogue writes quickgelu'(u) to aux_out");
      return launch_tn<4>(x, w, bias, y, aux_out, nullptr, nullptr, M, N, K, sched, st);
    case LVL_EPI_MUL_AUX_COLSUM: {
      LVL_REQUIRE(aux_in != nullptr && colsum != nullptr && ws != nullptr && lvl_aligned16(ws),
                  "linear_tn: the multiply-by-aux epilogue needs aux_in, colsum and a workspace");
      const int rc = launch_tn<5>(x, w, nullptr, y, nullptr, aux_in, ws, M, N, K, sched, st);
      if (rc != LVL_OK) return rc;
      const int P = (int)(2 * ((M + TM - 1) / TM));
      return lvl_launch_column_reduce(ws, P, N, N, ws + (size_t)P * N, colsum, nullptr, nullptr, st);
    }
    default:
      return lvl_fail(LVL_EINVAL, "linear_tn: unknown epilogue %d", epilogue);
  }
}
