// cave_hip.hip — gfx950 kernels and the C ABI declared in include/cave_hip.h.
//
// One 64-lane wavefront per training instance (workgroup = one wave, grid = B):
//   1. stream the instance's dense (m_max x d) block from HBM with 16-byte loads,
//      8 KiB in flight per wave, keeping only the non-zeros (ordered CSR in LDS);
//   2. classify rows / pair equalities / build CSC            (cone_core.h build_cone)
//   3. projected semismooth Newton in fp64, Hessian solve in registers (solve_cone)
//   4. fused epilogue: proj, rnorm, loss target, loss, d loss / d pred.
// Instances are independent, so the block->instance map is the identity and no
// XCD-aware remap is needed (nothing is shared through L2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/cave_hip.h"
#include "cone_common.h"
#include "cone_core.h"
#include "ctx_wave.h"
#include "cone_instance.h"

namespace cave {

#ifdef CAVE_STAMPS
__device__ unsigned long long g_stamp_buf[16 * 8192];
#endif

__global__ __launch_bounds__(64) void cone_dense_kernel(DenseParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WaveCtx c;
  c.lane = (int)threadIdx.x;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
#ifdef CAVE_STAMPS
    for (int i = 0; i < 16; ++i) c.st[i] = 0;
    unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    run_dense_instance(c, smem, P, b);
#ifdef CAVE_STAMPS
    c.st[14] = __builtin_amdgcn_s_memtime() - mt0;
    c.st[15] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz
    if (c.lane == 0 && b < 8192) for (int i = 0; i < 16; ++i) g_stamp_buf[b * 16 + i] = c.st[i];
#endif
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void cone_pack_kernel(PackParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WaveCtx c;
  c.lane = (int)threadIdx.x;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
    run_pack_instance(c, smem, P, b);
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void cone_packed_kernel(PackedParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WaveCtx c;
  c.lane = (int)threadIdx.x;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
    run_packed_instance(c, smem, P, b);
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
                            float* proj, float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                            int32_t* iters, void* stream) {
  if (B < 0 || m_max < 0 || d <= 0 || d > 65535) return fail(CAVE_E_INVALID, "cone_dense: bad shape (need 0 < d <= 65535)");
  if (m_max * d >= (int64_t)1 << 32) return fail(CAVE_E_INVALID, "cone_dense: m_max*d must be < 2^32");
  if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_dense: bad mode");
  if (B == 0) return CAVE_OK;
  if (!ctrs && m_max > 0) return fail(CAVE_E_INVALID, "cone_dense: ctrs is null");
  if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_dense: pred is null");
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes)) return fail(CAVE_E_INVALID, "cone_dense: bad nnz_cap / lds_bytes");
  hipError_t e = ensure_lds(cone_dense_kernel, (uint32_t)lds_bytes);
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(cone_dense_kernel)", e);
  DenseParams P;
  P.ctrs = ctrs; P.pred = pred; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d; P.mode = mode;
  P.sign = sign; P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : 100;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  unsigned grid = (unsigned)(B < (int64_t)1 << 30 ? B : (int64_t)1 << 30);
  hipLaunchKernelGGL(cone_dense_kernel, dim3(grid), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, P);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "launch cone_dense_kernel", e);
  return CAVE_OK;
}

int32_t cave_hip_pack_count(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int32_t nnz_cap, int32_t lds_bytes,
                            int32_t* n_rows, int32_t* n_nnz, int32_t* status, void* stream) {
  if (B < 0 || m_max < 0 || d <= 0 || d > 65535 || m_max * d >= (int64_t)1 << 32)
    return fail(CAVE_E_INVALID, "pack_count: bad shape");
  if (B == 0) return CAVE_OK;
  if (!ctrs || !n_rows || !n_nnz) return fail(CAVE_E_INVALID, "pack_count: null pointer");
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes)) return fail(CAVE_E_INVALID, "pack_count: bad limits");
  hipError_t e = ensure_lds(cone_pack_kernel, (uint32_t)lds_bytes);
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(cone_pack_kernel)", e);
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.n_rows = n_rows; P.n_nnz = n_nnz; P.status = status; P.fill = 0;
  hipLaunchKernelGGL(cone_pack_kernel, dim3((unsigned)B), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, P);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "launch cone_pack_kernel(count)", e);
  return CAVE_OK;
}

int32_t cave_hip_pack_fill(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int32_t nnz_cap, int32_t lds_bytes,
                           const cave_cone_store* store, int64_t slot0, int32_t* status, void* stream) {
  if (B < 0 || m_max < 0 || d <= 0 || d > 65535 || m_max * d >= (int64_t)1 << 32)
    return fail(CAVE_E_INVALID, "pack_fill: bad shape");
  if (B == 0) return CAVE_OK;
  if (!ctrs || !store) return fail(CAVE_E_INVALID, "pack_fill: null pointer");
  if (store->d != d || slot0 < 0 || slot0 + B > store->n) return fail(CAVE_E_INVALID, "pack_fill: store mismatch");
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes)) return fail(CAVE_E_INVALID, "pack_fill: bad limits");
  hipError_t e = ensure_lds(cone_pack_kernel, (uint32_t)lds_bytes);
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(cone_pack_kernel)", e);
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.status = status; P.store = *store; P.slot0 = slot0; P.fill = 1;
  hipLaunchKernelGGL(cone_pack_kernel, dim3((unsigned)B), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, P);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "launch cone_pack_kernel(fill)", e);
  return CAVE_OK;
}

int32_t cave_hip_packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz) {
  if (d <= 0 || max_rows < 0 || max_nnz < 0) return CAVE_E_INVALID;
  int32_t s = packed_lds_bytes(d, max_rows, max_nnz);
  return s < 0 ? CAVE_E_INVALID : s;
}

int32_t cave_hip_cone_packed(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                             int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                             float* proj, float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                             int32_t* iters, void* stream) {
  if (!store || B < 0) return fail(CAVE_E_INVALID, "cone_packed: null store / bad B");
  if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_packed: bad mode");
  if (B == 0) return CAVE_OK;
  if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_packed: pred is null");
  if (lds_bytes <= 0 || (uint32_t)lds_bytes > kMaxLds) return fail(CAVE_E_INVALID, "cone_packed: bad lds_bytes");
  hipError_t e = ensure_lds(cone_packed_kernel, (uint32_t)lds_bytes);
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "hipFuncSetAttribute(cone_packed_kernel)", e);
  PackedParams P;
  P.store = *store; P.ids = ids; P.pred = pred; P.B = B; P.mode = mode; P.sign = sign;
  P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : 100; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  hipLaunchKernelGGL(cone_packed_kernel, dim3((unsigned)B), dim3(64), (size_t)lds_bytes, (hipStream_t)stream, P);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "launch cone_packed_kernel", e);
  return CAVE_OK;
}

}  // extern "C"
