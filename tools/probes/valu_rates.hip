// Micro-benchmark: issue cost of single VALU / cross-lane instructions on gfx950, in shader cycles per wave64 instruction per
// SIMD, measured with 256 workgroups x 8 waves (two waves per SIMD) or x 16 waves (four per SIMD). Every kernel runs a loop
// of 8 x 8 independent instances of ONE instruction (inline asm, eight register chains), timed with s_memtime inside the
// kernel; the figure printed is elapsed_cycles / (instructions per wave x waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o /tmp/valu_rates ; run: /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(S) S S S S S S S S

// eight 32-bit chains a0..a7 (operands %0..%7), inputs %8 (b) and %9 (c)
#define K32(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                                       \
  __global__ void NAME(unsigned long long* out, int iters) {                                                            \
    float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,    \
          a6 = a0 + 6.f, a7 = a0 + 7.f;                                                                                 \
    const float b = 1.0001f, c = 0.5f;                                                                                  \
    __syncthreads();                                                                                                    \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                         \
    for (int it = 0; it < iters; ++it) {                                                                                \
      asm volatile(REP8(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n")                                \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
                   : "v"(b), "v"(c)                                                                                     \
                   : "vcc", "s20", "s21", "s22", "s23", "s24");                                                                                            \
    }                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                  \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                         \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                      \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;                                                \
  }
#define SAME8(OP) OP(0), OP(1), OP(2), OP(3), OP(4), OP(5), OP(6), OP(7)

// eight 64-bit chains (register pairs)
#define K64(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                                       \
  __global__ void NAME(unsigned long long* out, int iters) {                                                            \
    f2 a0 = {threadIdx.x * 1e-3f + 1.f, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, \
       a6 = a0 + 6.f, a7 = a0 + 7.f;                                                                                    \
    const f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};                                                                 \
    __syncthreads();                                                                                                    \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                         \
    for (int it = 0; it < iters; ++it) {                                                                                \
      asm volatile(REP8(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n")                                \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
                   : "v"(b), "v"(c)                                                                                     \
                   : "vcc", "s20", "s21", "s22", "s23", "s24");                                                                                            \
    }                                                                                                                   \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                         \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                      \
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                 \
    if (s.x + s.y == 12345.678f) out[0] = 0;                                                                            \
  }

#define I3(op, n) op " %" #n ", %" #n ", %8"
#define I3C(op, n) op " %" #n ", %" #n ", %8, %9"
#define I2(op, n) op " %" #n ", %" #n

K32(k_mul, I3("v_mul_f32", 0), I3("v_mul_f32", 1), I3("v_mul_f32", 2), I3("v_mul_f32", 3), I3("v_mul_f32", 4), I3("v_mul_f32", 5), I3("v_mul_f32", 6), I3("v_mul_f32", 7))
K32(k_add, I3("v_add_f32", 0), I3("v_add_f32", 1), I3("v_add_f32", 2), I3("v_add_f32", 3), I3("v_add_f32", 4), I3("v_add_f32", 5), I3("v_add_f32", 6), I3("v_add_f32", 7))
K32(k_max, I3("v_max_f32", 0), I3("v_max_f32", 1), I3("v_max_f32", 2), I3("v_max_f32", 3), I3("v_max_f32", 4), I3("v_max_f32", 5), I3("v_max_f32", 6), I3("v_max_f32", 7))
K32(k_fma, I3C("v_fma_f32", 0), I3C("v_fma_f32", 1), I3C("v_fma_f32", 2), I3C("v_fma_f32", 3), I3C("v_fma_f32", 4), I3C("v_fma_f32", 5), I3C("v_fma_f32", 6), I3C("v_fma_f32", 7))
K32(k_fmac, I3("v_fmac_f32", 0), I3("v_fmac_f32", 1), I3("v_fmac_f32", 2), I3("v_fmac_f32", 3), I3("v_fmac_f32", 4), I3("v_fmac_f32", 5), I3("v_fmac_f32", 6), I3("v_fmac_f32", 7))
K32(k_max3, I3C("v_max3_f32", 0), I3C("v_max3_f32", 1), I3C("v_max3_f32", 2), I3C("v_max3_f32", 3), I3C("v_max3_f32", 4), I3C("v_max3_f32", 5), I3C("v_max3_f32", 6), I3C("v_max3_f32", 7))
K32(k_exp, I2("v_exp_f32", 0), I2("v_exp_f32", 1), I2("v_exp_f32", 2), I2("v_exp_f32", 3), I2("v_exp_f32", 4), I2("v_exp_f32", 5), I2("v_exp_f32", 6), I2("v_exp_f32", 7))
K32(k_rcp, I2("v_rcp_f32", 0), I2("v_rcp_f32", 1), I2("v_rcp_f32", 2), I2("v_rcp_f32", 3), I2("v_rcp_f32", 4), I2("v_rcp_f32", 5), I2("v_rcp_f32", 6), I2("v_rcp_f32", 7))
K32(k_rsq, I2("v_rsq_f32", 0), I2("v_rsq_f32", 1), I2("v_rsq_f32", 2), I2("v_rsq_f32", 3), I2("v_rsq_f32", 4), I2("v_rsq_f32", 5), I2("v_rsq_f32", 6), I2("v_rsq_f32", 7))
K32(k_log, I2("v_log_f32", 0), I2("v_log_f32", 1), I2("v_log_f32", 2), I2("v_log_f32", 3), I2("v_log_f32", 4), I2("v_log_f32", 5), I2("v_log_f32", 6), I2("v_log_f32", 7))
K32(k_mov, I2("v_mov_b32", 0), I2("v_mov_b32", 1), I2("v_mov_b32", 2), I2("v_mov_b32", 3), I2("v_mov_b32", 4), I2("v_mov_b32", 5), I2("v_mov_b32", 6), I2("v_mov_b32", 7))
K32(k_cvtpk, I3("v_cvt_pk_bf16_f32", 0), I3("v_cvt_pk_bf16_f32", 1), I3("v_cvt_pk_bf16_f32", 2), I3("v_cvt_pk_bf16_f32", 3), I3("v_cvt_pk_bf16_f32", 4), I3("v_cvt_pk_bf16_f32", 5), I3("v_cvt_pk_bf16_f32", 6), I3("v_cvt_pk_bf16_f32", 7))
K32(k_lshl, "v_lshlrev_b32 %0, 16, %0", "v_lshlrev_b32 %1, 16, %1", "v_lshlrev_b32 %2, 16, %2", "v_lshlrev_b32 %3, 16, %3", "v_lshlrev_b32 %4, 16, %4", "v_lshlrev_b32 %5, 16, %5", "v_lshlrev_b32 %6, 16, %6", "v_lshlrev_b32 %7, 16, %7")
K32(k_and, I3("v_and_b32", 0), I3("v_and_b32", 1), I3("v_and_b32", 2), I3("v_and_b32", 3), I3("v_and_b32", 4), I3("v_and_b32", 5), I3("v_and_b32", 6), I3("v_and_b32", 7))
K32(k_addu, I3("v_add_u32", 0), I3("v_add_u32", 1), I3("v_add_u32", 2), I3("v_add_u32", 3), I3("v_add_u32", 4), I3("v_add_u32", 5), I3("v_add_u32", 6), I3("v_add_u32", 7))
K32(k_lshladd, I3C("v_lshl_add_u32", 0), I3C("v_lshl_add_u32", 1), I3C("v_lshl_add_u32", 2), I3C("v_lshl_add_u32", 3), I3C("v_lshl_add_u32", 4), I3C("v_lshl_add_u32", 5), I3C("v_lshl_add_u32", 6), I3C("v_lshl_add_u32", 7))
K32(k_mullo, I3("v_mul_lo_u32", 0), I3("v_mul_lo_u32", 1), I3("v_mul_lo_u32", 2), I3("v_mul_lo_u32", 3), I3("v_mul_lo_u32", 4), I3("v_mul_lo_u32", 5), I3("v_mul_lo_u32", 6), I3("v_mul_lo_u32", 7))
K32(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc", "v_cndmask_b32 %1, %1, %8, vcc", "v_cndmask_b32 %2, %2, %8, vcc", "v_cndmask_b32 %3, %3, %8, vcc", "v_cndmask_b32 %4, %4, %8, vcc", "v_cndmask_b32 %5, %5, %8, vcc", "v_cndmask_b32 %6, %6, %8, vcc", "v_cndmask_b32 %7, %7, %8, vcc")
K32(k_cmp, "v_cmp_gt_f32 vcc, %0, %8", "v_cmp_gt_f32 vcc, %1, %8", "v_cmp_gt_f32 vcc, %2, %8", "v_cmp_gt_f32 vcc, %3, %8", "v_cmp_gt_f32 vcc, %4, %8", "v_cmp_gt_f32 vcc, %5, %8", "v_cmp_gt_f32 vcc, %6, %8", "v_cmp_gt_f32 vcc, %7, %8")
#define DPP(n) "v_add_f32_dpp %" #n ", %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"
K32(k_dpp_add, DPP(0), DPP(1), DPP(2), DPP(3), DPP(4), DPP(5), DPP(6), DPP(7))
#define DPPM(n) "v_mov_b32_dpp %" #n ", %" #n " row_ror:8 row_mask:0xf bank_mask:0xf"
K32(k_dpp_mov, DPPM(0), DPPM(1), DPPM(2), DPPM(3), DPPM(4), DPPM(5), DPPM(6), DPPM(7))
K32(k_swap32, "v_permlane32_swap_b32 %0, %1", "v_permlane32_swap_b32 %2, %3", "v_permlane32_swap_b32 %4, %5", "v_permlane32_swap_b32 %6, %7", "v_permlane32_swap_b32 %0, %1", "v_permlane32_swap_b32 %2, %3", "v_permlane32_swap_b32 %4, %5", "v_permlane32_swap_b32 %6, %7")
K32(k_swap16, "v_permlane16_swap_b32 %0, %1", "v_permlane16_swap_b32 %2, %3", "v_permlane16_swap_b32 %4, %5", "v_permlane16_swap_b32 %6, %7", "v_permlane16_swap_b32 %0, %1", "v_permlane16_swap_b32 %2, %3", "v_permlane16_swap_b32 %4, %5", "v_permlane16_swap_b32 %6, %7")
#define BPERM(n) "ds_bpermute_b32 %" #n ", %8, %" #n
K32(k_bpermute, BPERM(0), BPERM(1), BPERM(2), BPERM(3), BPERM(4), BPERM(5), BPERM(6), BPERM(7) "\n s_waitcnt lgkmcnt(0)")
#define SWZ(n) "ds_swizzle_b32 %" #n ", %" #n " offset:swizzle(SWAP,16)"
K32(k_swizzle, SWZ(0), SWZ(1), SWZ(2), SWZ(3), SWZ(4), SWZ(5), SWZ(6), SWZ(7) "\n s_waitcnt lgkmcnt(0)")
K32(k_readlane, "v_readfirstlane_b32 s20, %0", "v_readfirstlane_b32 s21, %1", "v_readfirstlane_b32 s22, %2", "v_readfirstlane_b32 s23, %3", "v_readfirstlane_b32 s20, %4", "v_readfirstlane_b32 s21, %5", "v_readfirstlane_b32 s22, %6", "v_readfirstlane_b32 s23, %7")
K32(k_dot2, I3("v_dot2c_f32_bf16", 0), I3("v_dot2c_f32_bf16", 1), I3("v_dot2c_f32_bf16", 2), I3("v_dot2c_f32_bf16", 3), I3("v_dot2c_f32_bf16", 4), I3("v_dot2c_f32_bf16", 5), I3("v_dot2c_f32_bf16", 6), I3("v_dot2c_f32_bf16", 7))
K32(k_pkmul16, I3("v_pk_mul_f16", 0), I3("v_pk_mul_f16", 1), I3("v_pk_mul_f16", 2), I3("v_pk_mul_f16", 3), I3("v_pk_mul_f16", 4), I3("v_pk_mul_f16", 5), I3("v_pk_mul_f16", 6), I3("v_pk_mul_f16", 7))
K32(k_exp16, I2("v_exp_f16", 0), I2("v_exp_f16", 1), I2("v_exp_f16", 2), I2("v_exp_f16", 3), I2("v_exp_f16", 4), I2("v_exp_f16", 5), I2("v_exp_f16", 6), I2("v_exp_f16", 7))

// v_cndmask_b32 forms (round 6: the VOP2 form with VCC measured 12-19 cycles) and per-lane select alternatives
K32(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]", "v_cndmask_b32_e64 %1, %1, %8, s[20:21]", "v_cndmask_b32_e64 %2, %2, %8, s[20:21]", "v_cndmask_b32_e64 %3, %3, %8, s[20:21]", "v_cndmask_b32_e64 %4, %4, %8, s[20:21]", "v_cndmask_b32_e64 %5, %5, %8, s[20:21]", "v_cndmask_b32_e64 %6, %6, %8, s[20:21]", "v_cndmask_b32_e64 %7, %7, %8, s[20:21]")
K32(k_cndmask_c, "v_cndmask_b32 %0, 0, %8, vcc", "v_cndmask_b32 %1, 0, %8, vcc", "v_cndmask_b32 %2, 0, %8, vcc", "v_cndmask_b32 %3, 0, %8, vcc", "v_cndmask_b32 %4, 0, %8, vcc", "v_cndmask_b32 %5, 0, %8, vcc", "v_cndmask_b32 %6, 0, %8, vcc", "v_cndmask_b32 %7, 0, %8, vcc")
K32(k_cndmask_indep, "v_cndmask_b32 %0, %8, %9, vcc", "v_cndmask_b32 %1, %8, %9, vcc", "v_cndmask_b32 %2, %8, %9, vcc", "v_cndmask_b32 %3, %8, %9, vcc", "v_cndmask_b32 %4, %8, %9, vcc", "v_cndmask_b32 %5, %8, %9, vcc", "v_cndmask_b32 %6, %8, %9, vcc", "v_cndmask_b32 %7, %8, %9, vcc")
K32(k_cmp_cnd, "v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc", "v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc", "v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc", "v_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc", "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc", "v_cmp_gt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc", "v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc", "v_cmp_gt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc")
K32(k_bfi, I3C("v_bfi_b32", 0), I3C("v_bfi_b32", 1), I3C("v_bfi_b32", 2), I3C("v_bfi_b32", 3), I3C("v_bfi_b32", 4), I3C("v_bfi_b32", 5), I3C("v_bfi_b32", 6), I3C("v_bfi_b32", 7))
K32(k_min, I3("v_min_f32", 0), I3("v_min_f32", 1), I3("v_min_f32", 2), I3("v_min_f32", 3), I3("v_min_f32", 4), I3("v_min_f32", 5), I3("v_min_f32", 6), I3("v_min_f32", 7))
K32(k_med3, I3C("v_med3_f32", 0), I3C("v_med3_f32", 1), I3C("v_med3_f32", 2), I3C("v_med3_f32", 3), I3C("v_med3_f32", 4), I3C("v_med3_f32", 5), I3C("v_med3_f32", 6), I3C("v_med3_f32", 7))
K32(k_andor, I3C("v_and_or_b32", 0), I3C("v_and_or_b32", 1), I3C("v_and_or_b32", 2), I3C("v_and_or_b32", 3), I3C("v_and_or_b32", 4), I3C("v_and_or_b32", 5), I3C("v_and_or_b32", 6), I3C("v_and_or_b32", 7))
K32(k_sub, I3("v_sub_f32", 0), I3("v_sub_f32", 1), I3("v_sub_f32", 2), I3("v_sub_f32", 3), I3("v_sub_f32", 4), I3("v_sub_f32", 5), I3("v_sub_f32", 6), I3("v_sub_f32", 7))
K32(k_ldexp, I3("v_ldexp_f32", 0), I3("v_ldexp_f32", 1), I3("v_ldexp_f32", 2), I3("v_ldexp_f32", 3), I3("v_ldexp_f32", 4), I3("v_ldexp_f32", 5), I3("v_ldexp_f32", 6), I3("v_ldexp_f32", 7))
K32(k_perm, I3C("v_perm_b32", 0), I3C("v_perm_b32", 1), I3C("v_perm_b32", 2), I3C("v_perm_b32", 3), I3C("v_perm_b32", 4), I3C("v_perm_b32", 5), I3C("v_perm_b32", 6), I3C("v_perm_b32", 7))
K32(k_or, I3("v_or_b32", 0), I3("v_or_b32", 1), I3("v_or_b32", 2), I3("v_or_b32", 3), I3("v_or_b32", 4), I3("v_or_b32", 5), I3("v_or_b32", 6), I3("v_or_b32", 7))
K32(k_lshr, "v_lshrrev_b32 %0, 16, %0", "v_lshrrev_b32 %1, 16, %1", "v_lshrrev_b32 %2, 16, %2", "v_lshrrev_b32 %3, 16, %3", "v_lshrrev_b32 %4, 16, %4", "v_lshrrev_b32 %5, 16, %5", "v_lshrrev_b32 %6, 16, %6", "v_lshrrev_b32 %7, 16, %7")
K32(k_mulimm, "v_mul_f32 %0, 0x3fb8aa3b, %0", "v_mul_f32 %1, 0x3fb8aa3b, %1", "v_mul_f32 %2, 0x3fb8aa3b, %2", "v_mul_f32 %3, 0x3fb8aa3b, %3", "v_mul_f32 %4, 0x3fb8aa3b, %4", "v_mul_f32 %5, 0x3fb8aa3b, %5", "v_mul_f32 %6, 0x3fb8aa3b, %6", "v_mul_f32 %7, 0x3fb8aa3b, %7")
K32(k_muls, "v_mul_f32 %0, s20, %0", "v_mul_f32 %1, s20, %1", "v_mul_f32 %2, s20, %2", "v_mul_f32 %3, s20, %3", "v_mul_f32 %4, s20, %4", "v_mul_f32 %5, s20, %5", "v_mul_f32 %6, s20, %6", "v_mul_f32 %7, s20, %7")
K32(k_salu, "s_add_u32 s20, s20, s21", "s_add_u32 s22, s22, s21", "s_add_u32 s23, s23, s21", "s_add_u32 s24, s24, s21", "s_add_u32 s20, s20, s21", "s_add_u32 s22, s22, s21", "s_add_u32 s23, s23, s21", "s_add_u32 s24, s24, s21")

K64(k_pkmul, I3("v_pk_mul_f32", 0), I3("v_pk_mul_f32", 1), I3("v_pk_mul_f32", 2), I3("v_pk_mul_f32", 3), I3("v_pk_mul_f32", 4), I3("v_pk_mul_f32", 5), I3("v_pk_mul_f32", 6), I3("v_pk_mul_f32", 7))
K64(k_pkadd, I3("v_pk_add_f32", 0), I3("v_pk_add_f32", 1), I3("v_pk_add_f32", 2), I3("v_pk_add_f32", 3), I3("v_pk_add_f32", 4), I3("v_pk_add_f32", 5), I3("v_pk_add_f32", 6), I3("v_pk_add_f32", 7))
K64(k_pkfma, I3C("v_pk_fma_f32", 0), I3C("v_pk_fma_f32", 1), I3C("v_pk_fma_f32", 2), I3C("v_pk_fma_f32", 3), I3C("v_pk_fma_f32", 4), I3C("v_pk_fma_f32", 5), I3C("v_pk_fma_f32", 6), I3C("v_pk_fma_f32", 7))
K64(k_pkmov, I3("v_pk_mov_b32", 0), I3("v_pk_mov_b32", 1), I3("v_pk_mov_b32", 2), I3("v_pk_mov_b32", 3), I3("v_pk_mov_b32", 4), I3("v_pk_mov_b32", 5), I3("v_pk_mov_b32", 6), I3("v_pk_mov_b32", 7))
K64(k_mov64, I2("v_mov_b64", 0), I2("v_mov_b64", 1), I2("v_mov_b64", 2), I2("v_mov_b64", 3), I2("v_mov_b64", 4), I2("v_mov_b64", 5), I2("v_mov_b64", 6), I2("v_mov_b64", 7))
K64(k_lshl64, "v_lshlrev_b64 %0, 1, %0", "v_lshlrev_b64 %1, 1, %1", "v_lshlrev_b64 %2, 1, %2", "v_lshlrev_b64 %3, 1, %3", "v_lshlrev_b64 %4, 1, %4", "v_lshlrev_b64 %5, 1, %5", "v_lshlrev_b64 %6, 1, %6", "v_lshlrev_b64 %7, 1, %7")
K64(k_addf64, I3("v_add_f64", 0), I3("v_add_f64", 1), I3("v_add_f64", 2), I3("v_add_f64", 3), I3("v_add_f64", 4), I3("v_add_f64", 5), I3("v_add_f64", 6), I3("v_add_f64", 7))

typedef void (*kern_t)(unsigned long long*, int);
struct Entry { const char* name; kern_t k; };

int main() {
  unsigned long long* out;
  hipMalloc(&out, 256 * 16 * 8);
  const Entry es[] = {
      {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_max_f32", k_max}, {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac},
      {"v_max3_f32", k_max3}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp}, {"v_rsq_f32", k_rsq}, {"v_log_f32", k_log},
      {"v_mov_b32", k_mov}, {"v_cvt_pk_bf16_f32", k_cvtpk}, {"v_lshlrev_b32", k_lshl}, {"v_and_b32", k_and},
      {"v_add_u32", k_addu}, {"v_lshl_add_u32", k_lshladd}, {"v_mul_lo_u32", k_mullo}, {"v_cndmask_b32", k_cndmask},
      {"v_cmp_gt_f32", k_cmp}, {"v_add_f32 dpp row_shr:1", k_dpp_add}, {"v_mov_b32 dpp row_ror:8", k_dpp_mov},
      {"v_permlane32_swap_b32", k_swap32}, {"v_permlane16_swap_b32", k_swap16}, {"ds_bpermute_b32 (+wait per 8)", k_bpermute},
      {"ds_swizzle_b32 SWAP,16 (+wait per 8)", k_swizzle}, {"v_readfirstlane_b32", k_readlane}, {"v_dot2c_f32_bf16", k_dot2},
      {"v_pk_mul_f16", k_pkmul16}, {"v_exp_f16", k_exp16}, {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd},
      {"v_pk_fma_f32", k_pkfma}, {"v_pk_mov_b32", k_pkmov}, {"v_mov_b64", k_mov64}, {"v_lshlrev_b64", k_lshl64},
      {"v_add_f64", k_addf64},
      {"v_cndmask_b32_e64 (SGPR pair mask)", k_cndmask64}, {"v_cndmask_b32 v, 0, v, vcc", k_cndmask_c},
      {"v_cndmask_b32 v, a, b, vcc (no chain)", k_cndmask_indep}, {"v_cmp_gt_f32 + v_cndmask_b32 (pair)", k_cmp_cnd},
      {"v_bfi_b32", k_bfi}, {"v_min_f32", k_min}, {"v_med3_f32", k_med3}, {"v_and_or_b32", k_andor}, {"v_sub_f32", k_sub},
      {"v_ldexp_f32", k_ldexp}, {"v_perm_b32", k_perm}, {"v_or_b32", k_or}, {"v_lshrrev_b32", k_lshr},
      {"v_mul_f32 v, literal, v", k_mulimm}, {"v_mul_f32 v, sgpr, v", k_muls}, {"s_add_u32", k_salu}};
  const int iters = 2000;
  printf("%-40s %12s %12s %12s\n", "instruction (wave64)", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
  for (const Entry& e : es) {
    printf("%-40s", e.name);
    for (int waves : {4, 8, 16}) {          // per workgroup = per CU: 1, 2, 4 per SIMD
      hipLaunchKernelGGL(e.k, dim3(256), dim3(64 * waves), 0, 0, out, 10);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(e.k, dim3(256), dim3(64 * waves), 0, 0, out, iters);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * waves);
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (auto v : h) s += (double)v;
      const double per_wave = s / h.size() / ((double)iters * 64);      // cycles per instruction as one wave sees them
      printf(" %12.2f", per_wave / (waves / 4));                        // per SIMD
    }
    printf("\n");
  }
  return 0;
}
