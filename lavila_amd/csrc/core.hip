// Library identification, error text and workspace sizing (host-only code).
#include <stdarg.h>

#include "common.h"

thread_local char lvl_err_buf[512] = "";

int lvl_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(lvl_err_buf, sizeof(lvl_err_buf), fmt, ap);
  va_end(ap);
  return code;
}

int lvl_ln_bwd_parts();
int lvl_gelu_bwd_row_blocks();
int lvl_qkv_bias_row_blocks();
int lvl_colsum_mid_rows();
int64_t lvl_wgrad_workspace_floats(int64_t N, int64_t K);
int64_t lvl_linear_tn_workspace_floats(int64_t M, int64_t N);

// Compute units the persistent kernels (lvl_linear_tn, lvl_linear_wgrad) size their grids for; 0 = all of the device.
static std::atomic<int> g_cu_limit{0};

int lvl_persistent_cus() {
  static std::atomic<int> cached[64];
  int dev = 0, v = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    v = cached[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
      cached[dev & 63].store(v, std::memory_order_relaxed);
    }
  }
  const int lim = g_cu_limit.load(std::memory_order_relaxed);
  return (lim > 0 && lim < v) ? lim : v;
}

extern "C" int lvl_set_compute_units(int n) {
  if (n < 0 || (n != 0 && (n < 8 || n % 8))) return lvl_fail(LVL_EINVAL, "set_compute_units: %d is not 0 or a multiple of 8", n);
  g_cu_limit.store(n, std::memory_order_relaxed);
  return LVL_OK;
}

// Test hook of the dynamic schedules: workgroups with blockIdx % mod == 1 act as if their compute unit had been held by
// another kernel for the whole launch (0 = off).
static std::atomic<int> g_late_mod{0};
int lvl_debug_late_mod() { return g_late_mod.load(std::memory_order_relaxed); }
extern "C" int lvl_debug_late_workgroups(int mod) {
  if (mod < 0 || mod == 1) return lvl_fail(LVL_EINVAL, "debug_late_workgroups: mod must be 0 or >= 2");
  g_late_mod.store(mod, std::memory_order_relaxed);
  return LVL_OK;
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (((reinterpret_cast<uintptr_t>(p) & 15) | (n & 3)) == 0) {
    float4* p4 = reinterpret_cast<float4*>(p);
    for (const size_t n4 = n >> 2; i < n4; i += stride) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (; i < n; i += stride) p[i] = 0.f;
  }
}

int lvl_zero_f32(float* p, size_t n, hipStream_t st) {
  if (n == 0) return LVL_OK;
  const size_t want = (n / 4 + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(zero_f32_kernel, dim3(blocks), dim3(256), 0, st, p, n);
  LVL_CHECK_LAUNCH("zero_f32");
  return LVL_OK;
}

extern "C" const char* lvl_version(void) { return "lavila_hip 0.1 (gfx950)"; }
extern "C" const char* lvl_last_error(void) { return lvl_err_buf; }

extern "C" int64_t lvl_workspace_floats(const char* op, int64_t rows, int64_t cols) {
  if (!op) return -1;
  // partial slabs + the kColsumMid intermediate rows of the two-stage column reduction behind them
  if (!strcmp(op, "layernorm_bwd")) return (int64_t)(lvl_ln_bwd_parts() + lvl_colsum_mid_rows()) * 3 * cols;
  if (!strcmp(op, "bias_quickgelu_bwd")) return (int64_t)(lvl_gelu_bwd_row_blocks() + lvl_colsum_mid_rows()) * cols;
  if (!strcmp(op, "divided_attn_fwd")) return rows * 64 * 66;   // <= 64 CLS-row partial records per (b,h)
  // delta [B*H, T] + the cls token's partial gradient records: <= 64 slots of [192] per (b, h) (one per frame / per
  // location chunk, added up in order by cls_grad_finalize_kernel)
  if (!strcmp(op, "divided_attn_bwd")) return rows * cols + rows * 192 * 64;
  if (!strcmp(op, "causal_attn_bwd")) return rows * cols;    // delta [B*H, L]
  if (!strcmp(op, "qkv_bias_grad")) return (int64_t)(lvl_qkv_bias_row_blocks() + lvl_colsum_mid_rows()) * 2 * cols;   // cols = D
  if (!strcmp(op, "linear_wgrad")) return lvl_wgrad_workspace_floats(rows, cols);   // rows = N (out), cols = K (in); -1: no tiling
  if (!strcmp(op, "linear_tn")) return lvl_linear_tn_workspace_floats(rows, cols);   // rows = M, cols = N
  return -1;
}
