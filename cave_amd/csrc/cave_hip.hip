// cave_hip.hip — gfx950 kernels and the C ABI declared in include/cave_hip.h.
//
// One workgroup of 1, 2 or 4 cooperating 64-lane wavefronts per training instance (grid = B; the
// `waves` argument of the C ABI picks the shape, see include/cave_hip.h):
//   1. stream the instance's dense (m_max x d) block from HBM with 16-byte loads, two batches in
//      flight per wave, keeping only the non-zeros (ordered CSR in LDS);
//   2. classify rows / pair equalities / build CSC            (cone_core.h build_cone)
//   3. projected semismooth Newton in fp64, Newton systems solved in registers (solve_cone)
//   4. fused epilogue: proj, rnorm, loss target, loss, d loss / d pred.
// Cones beyond LDS run on persistent 4-wave workgroups over a global workspace (the *_large kernels).
// Instances are independent, so the block->instance map is the identity and no
// XCD-aware remap is needed (nothing is shared through L2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/cave_hip.h"
#include "cone_common.h"
#include "cone_core.h"
#include "ctx_wave.h"
#include "ctx_block.h"
#include "cone_instance.h"

namespace cave {

#ifdef CAVE_STAMPS
__device__ unsigned long long g_stamp_buf[16 * 8192];
#endif

using Ctx1 = WaveCtx;      // one wave per instance, reduced systems up to 64 rows
using Ctx4 = BlockCtx<4>;  // 4-wave workgroup per instance, reduced systems up to 32 rows
using Ctx2 = BlockCtx<2>;  // 2-wave workgroup per instance
using CtxW = BlockCtx<4, true>;  // 4 waves with the full register budget: for launches whose LDS arena allows one
                                 // workgroup per CU anyway (TSP-50: 100-160 KB); reduced systems up to 64 rows

// launch bounds: NT threads; for the 4-wave context ask for 4 waves per SIMD (= 4 workgroups per CU,
// the residency LDS allows), which caps the kernel at 128 VGPRs
#define CAVE_BOUNDS(C) __launch_bounds__(C::NT, C::MIN_WAVES_PER_EU)

template <class C>
__global__ CAVE_BOUNDS(C) void cone_dense_kernel(DenseParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  // one workgroup per instance (grid = B): no loop, so nothing loop-invariant is hoisted into long live ranges
  const int64_t b = blockIdx.x;
  if (b >= P.B) return;
#ifdef CAVE_STAMPS
  for (int i = 0; i < 32; ++i) c.st[i] = 0;
  unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  run_dense_instance(c, smem, P, b);
#ifdef CAVE_STAMPS
  c.st[14] = __builtin_amdgcn_s_memtime() - mt0;
  c.st[15] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz
  if (c.tid() == 0 && b < 8192) for (int i = 0; i < 16; ++i) g_stamp_buf[b * 16 + i] = c.st[i];
#endif
}

template <class C>
__global__ CAVE_BOUNDS(C) void cone_pack_kernel(PackParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  const int64_t b = blockIdx.x;
  if (b < P.B) run_pack_instance(c, smem, P, b);
}

template <class C>
__global__ CAVE_BOUNDS(C) void cone_packed_kernel(PackedParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  const int64_t b = blockIdx.x;
  if (b < P.B) run_packed_instance(c, smem, P, b);
}

// ---- large-cone path (cone_band.h): persistent 4-wave workgroups, arena = a slice of a global
// workspace, LDS = "hot" arena for the small per-iteration arrays.  256 VGPRs (2 waves / SIMD).
struct LargeWs {
  unsigned char* base;
  uint64_t slice;  // bytes per workgroup (< 4 GiB)
};
using CtxL = BlockCtx<4, true>;
using CtxL2 = BlockCtx<2, true>;

__global__ __launch_bounds__(CtxL::NT, 2) void cone_dense_large_kernel(DenseParams P, LargeWs W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CtxL c;
  c.init(smem);
  unsigned char* ws = W.base + (uint64_t)blockIdx.x * W.slice;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
    run_dense_instance<CtxL, true>(c, smem, P, b, ws, (uint32_t)W.slice);
    __syncthreads();
  }
}

__global__ __launch_bounds__(CtxL::NT, 2) void cone_pack_large_kernel(PackParams P, LargeWs W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  CtxL c;
  c.init(smem);
  unsigned char* ws = W.base + (uint64_t)blockIdx.x * W.slice;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
    run_pack_instance<CtxL, true>(c, smem, P, b, ws, (uint32_t)W.slice);
    __syncthreads();
  }
}

// (MINW = waves per SIMD the register budget is set for: 2 -> 256 VGPRs.  C = CtxL: 4 waves, two workgroups per CU;
//  CtxL2 / Ctx1: 2 / 1 waves, four and more workgroups per CU where the LDS allows -- for batches that fill the
//  chip several times over with narrow-band cones, whose elimination runs on one wave anyway: cone_band.h)
template <class C, int MINB>
__global__ __launch_bounds__(C::NT, MINB) void cone_packed_large_kernel(PackedParams P, LargeWs W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  unsigned char* ws = W.base + (uint64_t)blockIdx.x * W.slice;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
#ifdef CAVE_STAMPS
    for (int i = 0; i < 32; ++i) c.st[i] = 0;
    unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    run_packed_large_instance<C>(c, smem, P, b, ws, (uint32_t)W.slice);
#ifdef CAVE_STAMPS
    c.st[14] = __builtin_amdgcn_s_memtime() - mt0;
    c.st[15] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz
    if (c.tid() == 0 && b < 8192) for (int i = 0; i < 16; ++i) g_stamp_buf[b * 16 + i] = c.st[i];
#endif
    __syncthreads();
  }
}

// --------------------------------------------------------------- host helpers

static thread_local char g_err[512] = "";

static int32_t fail(int32_t code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess) snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  else snprintf(g_err, sizeof(g_err), "%s", what);
  return code;
}

template <class K>
static hipError_t ensure_lds(K kernel, uint32_t bytes) {
  if (bytes <= 48u * 1024u) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

// launch kernel<Ctx4> or kernel<Ctx1> on B workgroups
#define CAVE_LAUNCH(KERNEL, WAVES, B, LDS, STREAM, PARAMS, WHAT)                                              \
  do {                                                                                                         \
    hipError_t e_;                                                                                             \
    if ((B) >= (int64_t)1 << 31) return fail(CAVE_E_INVALID, WHAT ": batch too large (B < 2^31)");                \
    unsigned grid_ = (unsigned)(B);                                                                            \
    if ((WAVES) == 1) {                                                                                        \
      e_ = ensure_lds(KERNEL<Ctx1>, (uint32_t)(LDS));                                                          \
      if (e_ != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(" WHAT ")", e_);                   \
      hipLaunchKernelGGL(KERNEL<Ctx1>, dim3(grid_), dim3(Ctx1::NT), (size_t)(LDS), (hipStream_t)(STREAM), PARAMS); \
    } else if ((WAVES) == 8) {                                                                                 \
      e_ = ensure_lds(KERNEL<CtxW>, (uint32_t)(LDS));                                                          \
      if (e_ != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(" WHAT ")", e_);                   \
      hipLaunchKernelGGL(KERNEL<CtxW>, dim3(grid_), dim3(CtxW::NT), (size_t)(LDS), (hipStream_t)(STREAM), PARAMS); \
    } else if ((WAVES) == 2) {                                                                                 \
      e_ = ensure_lds(KERNEL<Ctx2>, (uint32_t)(LDS));                                                          \
      if (e_ != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(" WHAT ")", e_);                   \
      hipLaunchKernelGGL(KERNEL<Ctx2>, dim3(grid_), dim3(Ctx2::NT), (size_t)(LDS), (hipStream_t)(STREAM), PARAMS); \
    } else {                                                                                                   \
      e_ = ensure_lds(KERNEL<Ctx4>, (uint32_t)(LDS));                                                          \
      if (e_ != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(" WHAT ")", e_);                   \
      hipLaunchKernelGGL(KERNEL<Ctx4>, dim3(grid_), dim3(Ctx4::NT), (size_t)(LDS), (hipStream_t)(STREAM), PARAMS); \
    }                                                                                                          \
    e_ = hipGetLastError();                                                                                    \
    if (e_ != hipSuccess) return fail(CAVE_E_LAUNCH, "launch " WHAT, e_);                                      \
  } while (0)

static bool waves_ok(int32_t& waves) {
  if (waves == 0) waves = 2;  // measured best at the benchmark size, no register spills, 64-row systems
  return waves == 1 || waves == 2 || waves == 4 || waves == 8;  // 8 = four waves, wide register budget
}

}  // namespace cave

using namespace cave;

extern "C" {

int32_t cave_hip_version(void) { return CAVE_HIP_ABI_VERSION; }

#ifdef CAVE_STAMPS
// diagnostic build only: read and clear the per-phase cycle accumulators
int32_t cave_hip_debug_stamps(unsigned long long* out, int n_inst) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(cave::g_stamp_buf), sizeof(unsigned long long) * 16 * (size_t)n_inst);
  return 0;
}
#endif

const char* cave_hip_last_error(void) { return g_err; }

int32_t cave_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t cave_hip_default_limits(int64_t m_max, int64_t d, int32_t* nnz_cap, int32_t* lds_bytes) {
  if (m_max < 0 || d <= 0) return fail(CAVE_E_INVALID, "default_limits: bad shape");
  default_limits(m_max, d, nnz_cap, lds_bytes);
  return CAVE_OK;
}

int32_t cave_hip_cone_dense(const float* ctrs, const float* pred, int64_t B, int64_t m_max, int64_t d, int32_t mode,
                            float sign, float inner_ratio, int32_t max_iter, int32_t nnz_cap, int32_t lds_bytes,
                            int32_t waves, float* proj, float* rnorm, float* target, float* loss, float* grad,
                            int32_t* status, int32_t* iters, void* stream) {
  if (B < 0 || m_max < 0 || d <= 0 || d > 65535) return fail(CAVE_E_INVALID, "cone_dense: bad shape (need 0 < d <= 65535)");
  if (m_max * d >= (int64_t)1 << 32) return fail(CAVE_E_INVALID, "cone_dense: m_max*d must be < 2^32");
  if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_INNER_IPM) return fail(CAVE_E_INVALID, "cone_dense: bad mode");
  if (B == 0) return CAVE_OK;
  if (!ctrs && m_max > 0) return fail(CAVE_E_INVALID, "cone_dense: ctrs is null");
  if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_dense: pred is null");
  if (!waves_ok(waves)) return fail(CAVE_E_INVALID, "cone_dense: waves must be 0, 1, 2, 4 or 8");
  // the one-wave lite solver only exists in the one- / two-wave shapes and only runs for grids of up to 2048
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes, waves <= 2 && B <= 2048))
    return fail(CAVE_E_INVALID, "cone_dense: bad nnz_cap / lds_bytes");
  DenseParams P;
  P.ctrs = ctrs; P.pred = pred; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d; P.mode = mode;
  P.sign = sign; P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100);
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  CAVE_LAUNCH(cone_dense_kernel, waves, B, lds_bytes, stream, P, "cone_dense_kernel");
  return CAVE_OK;
}

int32_t cave_hip_pack_count(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int32_t nnz_cap, int32_t lds_bytes,
                            int32_t waves, int32_t* n_rows, int32_t* n_nnz, int32_t* status, void* stream) {
  if (B < 0 || m_max < 0 || d <= 0 || d > 65535 || m_max * d >= (int64_t)1 << 32)
    return fail(CAVE_E_INVALID, "pack_count: bad shape");
  if (B == 0) return CAVE_OK;
  if (!ctrs || !n_rows || !n_nnz) return fail(CAVE_E_INVALID, "pack_count: null pointer");
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes, false)) return fail(CAVE_E_INVALID, "pack_count: bad limits");
  if (!waves_ok(waves)) return fail(CAVE_E_INVALID, "pack_count: waves must be 0, 1, 2, 4 or 8");
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.n_rows = n_rows; P.n_nnz = n_nnz; P.status = status; P.fill = 0;
  CAVE_LAUNCH(cone_pack_kernel, waves, B, lds_bytes, stream, P, "cone_pack_kernel(count)");
  return CAVE_OK;
}

int32_t cave_hip_pack_fill(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int32_t nnz_cap, int32_t lds_bytes,
                           int32_t waves, const cave_cone_store* store, int64_t slot0, int32_t* status, void* stream) {
  if (B < 0 || m_max < 0 || d <= 0 || d > 65535 || m_max * d >= (int64_t)1 << 32)
    return fail(CAVE_E_INVALID, "pack_fill: bad shape");
  if (B == 0) return CAVE_OK;
  if (!ctrs || !store) return fail(CAVE_E_INVALID, "pack_fill: null pointer");
  if (store->d != d || slot0 < 0 || slot0 + B > store->n) return fail(CAVE_E_INVALID, "pack_fill: store mismatch");
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes, false)) return fail(CAVE_E_INVALID, "pack_fill: bad limits");
  if (!waves_ok(waves)) return fail(CAVE_E_INVALID, "pack_fill: waves must be 0, 1, 2, 4 or 8");
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  if ((store->n_rows == nullptr) != (store->n_nnz == nullptr)) return fail(CAVE_E_INVALID, "pack_fill: n_rows and n_nnz go together");
  P.status = status; P.store = *store; P.slot0 = slot0; P.fill = 1;
  CAVE_LAUNCH(cone_pack_kernel, waves, B, lds_bytes, stream, P, "cone_pack_kernel(fill)");
  return CAVE_OK;
}

int32_t cave_hip_packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz, int32_t all_pm1) {
  if (d <= 0 || max_rows < 0 || max_nnz < 0) return CAVE_E_INVALID;
  int32_t s = packed_lds_bytes(d, max_rows, max_nnz, all_pm1 != 0, all_pm1 != 2);
  return s < 0 ? CAVE_E_INVALID : s;
}

int32_t cave_hip_cone_packed(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                             int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                             int32_t waves, float* proj, float* rnorm, float* target, float* loss, float* grad,
                             int32_t* status, int32_t* iters, void* stream) {
  if (!store || B < 0) return fail(CAVE_E_INVALID, "cone_packed: null store / bad B");
  if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_INNER_IPM) return fail(CAVE_E_INVALID, "cone_packed: bad mode");
  if (B == 0) return CAVE_OK;
  if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_packed: pred is null");
  if (lds_bytes <= 0 || (uint32_t)lds_bytes > kMaxLds) return fail(CAVE_E_INVALID, "cone_packed: bad lds_bytes");
  if (!waves_ok(waves)) return fail(CAVE_E_INVALID, "cone_packed: waves must be 0, 1, 2, 4 or 8");
  PackedParams P;
  P.store = *store; P.ids = ids; P.pred = pred; P.B = B; P.mode = mode; P.sign = sign;
  P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100);
  P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  CAVE_LAUNCH(cone_packed_kernel, waves, B, lds_bytes, stream, P, "cone_packed_kernel");
  return CAVE_OK;
}

// ------------------------------------------------------------------ large-cone path

int64_t cave_hip_large_slice_bytes(int64_t m_max, int64_t d, int64_t nnz_cap, int64_t band_entries) {
  if (m_max < 0 || d <= 0 || nnz_cap <= 0 || band_entries < 0) return CAVE_E_INVALID;
  return (int64_t)large_slice_bytes(m_max, d, nnz_cap, band_entries);
}

int64_t cave_hip_packed_large_slice_bytes(int64_t d, int64_t max_rows, int64_t band_entries) {
  if (d <= 0 || max_rows < 0 || band_entries < 0) return CAVE_E_INVALID;
  return (int64_t)packed_large_slice_bytes(d, max_rows, band_entries);
}

int32_t cave_hip_packed_large_lds_bytes(int32_t max_rows, int32_t max_bw) {
  if (max_rows < 0 || max_bw < 0) return CAVE_E_INVALID;
  return (int32_t)packed_large_lds_bytes(max_rows, max_bw);
}

static int32_t check_large(const char* who, const void* workspace, int64_t slice_bytes, int32_t n_slots, int32_t& lds_bytes) {
  if (!workspace || slice_bytes < 1024 || slice_bytes >= ((int64_t)1 << 32) || (slice_bytes & 7) || n_slots <= 0 ||
      ((uintptr_t)workspace & 15u)) {
    snprintf(g_err, sizeof(g_err), "%s: bad workspace (16-byte aligned, 1 KiB <= slice_bytes < 4 GiB, multiple of 8, n_slots > 0)", who);
    return CAVE_E_INVALID;
  }
  if (lds_bytes <= 0) lds_bytes = 64 * 1024;
  if (lds_bytes < 1024 || (uint32_t)lds_bytes > kMaxLds) {
    snprintf(g_err, sizeof(g_err), "%s: bad lds_bytes", who);
    return CAVE_E_INVALID;
  }
  return CAVE_OK;
}

#define CAVE_LAUNCH_LARGE(KERNEL, B, SLOTS, LDS, STREAM, PARAMS, WS, WHAT) CAVE_LAUNCH_LARGE_NT(KERNEL, CtxL::NT, B, SLOTS, LDS, STREAM, PARAMS, WS, WHAT)
#define CAVE_LAUNCH_LARGE_NT(KERNEL, NT_, B, SLOTS, LDS, STREAM, PARAMS, WS, WHAT)                             \
  do {                                                                                                         \
    hipError_t e_ = ensure_lds(KERNEL, (uint32_t)(LDS));                                                       \
    if (e_ != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(" WHAT ")", e_);                     \
    unsigned grid_ = (unsigned)((B) < (int64_t)(SLOTS) ? (B) : (int64_t)(SLOTS));                              \
    hipLaunchKernelGGL(KERNEL, dim3(grid_), dim3(NT_), (size_t)(LDS), (hipStream_t)(STREAM), PARAMS, WS); \
    e_ = hipGetLastError();                                                                                    \
    if (e_ != hipSuccess) return fail(CAVE_E_LAUNCH, "launch " WHAT, e_);                                      \
  } while (0)

int32_t cave_hip_cone_dense_large(const float* ctrs, const float* pred, int64_t B, int64_t m_max, int64_t d,
                                  int32_t mode, float sign, float inner_ratio, int32_t max_iter, int64_t nnz_cap,
                                  int32_t lds_bytes, void* workspace, int64_t slice_bytes, int32_t n_slots, float* proj,
                                  float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                                  int32_t* iters, void* stream) {
  if (B < 0 || m_max < 0 || m_max > 65535 || d <= 0 || d > 65535 || m_max * d >= (int64_t)1 << 32)
    return fail(CAVE_E_INVALID, "cone_dense_large: bad shape (need m_max, d <= 65535, m_max*d < 2^32)");
  if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_dense_large: bad mode");
  if (B == 0) return CAVE_OK;
  if (!ctrs && m_max > 0) return fail(CAVE_E_INVALID, "cone_dense_large: ctrs is null");
  if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_dense_large: pred is null");
  if (nnz_cap <= 0 || nnz_cap >= (int64_t)1 << 31) return fail(CAVE_E_INVALID, "cone_dense_large: bad nnz_cap");
  int32_t rc = check_large("cone_dense_large", workspace, slice_bytes, n_slots, lds_bytes);
  if (rc != CAVE_OK) return rc;
  DenseParams P;
  P.ctrs = ctrs; P.pred = pred; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d; P.mode = mode;
  P.sign = sign; P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : 100;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  LargeWs W{(unsigned char*)workspace, (uint64_t)slice_bytes};
  CAVE_LAUNCH_LARGE(cone_dense_large_kernel, B, n_slots, lds_bytes, stream, P, W, "cone_dense_large_kernel");
  return CAVE_OK;
}

int32_t cave_hip_pack_large(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int64_t nnz_cap, void* workspace,
                            int64_t slice_bytes, int32_t n_slots, int32_t* n_rows, int32_t* n_nnz,
                            const cave_cone_store* store, int64_t slot0, int32_t* status, void* stream) {
  if (B < 0 || m_max < 0 || m_max > 65535 || d <= 0 || d > 65535 || m_max * d >= (int64_t)1 << 32)
    return fail(CAVE_E_INVALID, "pack_large: bad shape");
  if (B == 0) return CAVE_OK;
  if (!ctrs) return fail(CAVE_E_INVALID, "pack_large: ctrs is null");
  if (!store && (!n_rows || !n_nnz)) return fail(CAVE_E_INVALID, "pack_large: count pass needs n_rows and n_nnz");
  if (store && (store->d != d || slot0 < 0 || slot0 + B > store->n)) return fail(CAVE_E_INVALID, "pack_large: store mismatch");
  if (nnz_cap <= 0 || nnz_cap >= (int64_t)1 << 31) return fail(CAVE_E_INVALID, "pack_large: bad nnz_cap");
  int32_t lds = 1024;
  int32_t rc = check_large("pack_large", workspace, slice_bytes, n_slots, lds);
  if (rc != CAVE_OK) return rc;
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds;
  P.n_rows = n_rows; P.n_nnz = n_nnz; P.status = status;
  if (store) { P.store = *store; P.slot0 = slot0; P.fill = 1; }
  LargeWs W{(unsigned char*)workspace, (uint64_t)slice_bytes};
  CAVE_LAUNCH_LARGE(cone_pack_large_kernel, B, n_slots, lds, stream, P, W, "cone_pack_large_kernel");
  return CAVE_OK;
}

int32_t cave_hip_cone_packed_large(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                                   int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                                   void* workspace, int64_t slice_bytes, int32_t n_slots, float* proj, float* rnorm,
                                   float* target, float* loss, float* grad, int32_t* status, int32_t* iters,
                                   void* stream) {
  if (!store || B < 0) return fail(CAVE_E_INVALID, "cone_packed_large: null store / bad B");
  if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_packed_large: bad mode");
  if (B == 0) return CAVE_OK;
  if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_packed_large: pred is null");
  int32_t rc = check_large("cone_packed_large", workspace, slice_bytes, n_slots, lds_bytes);
  if (rc != CAVE_OK) return rc;
  PackedParams P;
  P.store = *store; P.ids = ids; P.pred = pred; P.B = B; P.mode = mode; P.sign = sign;
  P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : 100; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  LargeWs W{(unsigned char*)workspace, (uint64_t)slice_bytes};
  // workgroup shape: 4 waves (two workgroups per CU) unless the batch needs more than two workgroups per CU and
  // four fit the LDS; then 2 waves.  CAVE_LARGE_WAVES=1|2|4 overrides (diagnostic).
  const int64_t grid = B < n_slots ? B : n_slots;
  int waves = (grid > 2 * 256 && (int64_t)lds_bytes * 4 <= (int64_t)kMaxLds) ? 2 : 4;
  if (const char* e = getenv("CAVE_LARGE_WAVES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) waves = v; }
  if (waves == 1) {
    CAVE_LAUNCH_LARGE_NT((cone_packed_large_kernel<Ctx1, 2>), Ctx1::NT, B, n_slots, lds_bytes, stream, P, W, "cone_packed_large_kernel<1>");
  } else if (waves == 2) {
    CAVE_LAUNCH_LARGE_NT((cone_packed_large_kernel<CtxL2, 2>), CtxL2::NT, B, n_slots, lds_bytes, stream, P, W, "cone_packed_large_kernel<2>");
  } else {
    CAVE_LAUNCH_LARGE_NT((cone_packed_large_kernel<CtxL, 2>), CtxL::NT, B, n_slots, lds_bytes, stream, P, W, "cone_packed_large_kernel<4>");
  }
  return CAVE_OK;
}

}  // extern "C"
