// cave_hip.hip — the C ABI declared in include/cave_hip.h (host code only).
//
// The kernels live in kernels.h; each shape is instantiated in its own k_*.hip next to its launch
// function.  This file validates arguments, fills the parameter blocks and picks the launch shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"

namespace cave {

// --------------------------------------------------------------- host helpers

static thread_local char g_err[512] = "";

static int32_t fail(int32_t code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess) snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  else snprintf(g_err, sizeof(g_err), "%s", what);
  return code;
}

// compute units of the current device (queried once per device; 256 on MI355X)
static int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// launch the `waves` shape of an LDS-path operator on B workgroups
#define CAVE_LAUNCH(OP, WAVES, B, LDS, STREAM, PARAMS, WHAT)                                                   \
  do {                                                                                                         \
    if ((B) >= (int64_t)1 << 31) return fail(CAVE_E_INVALID, WHAT ": batch too large (B < 2^31)");                \
    const unsigned grid_ = (unsigned)(B);                                                                      \
    hipError_t e_;                                                                                             \
    if ((WAVES) == 1) e_ = launch_##OP##_w1(grid_, (uint32_t)(LDS), (hipStream_t)(STREAM), PARAMS);            \
    else if ((WAVES) == 8) e_ = launch_##OP##_w8(grid_, (uint32_t)(LDS), (hipStream_t)(STREAM), PARAMS);       \
    else if ((WAVES) == 2) e_ = launch_##OP##_w2(grid_, (uint32_t)(LDS), (hipStream_t)(STREAM), PARAMS);       \
    else e_ = launch_##OP##_w4(grid_, (uint32_t)(LDS), (hipStream_t)(STREAM), PARAMS);                         \
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
  CAVE_LAUNCH(dense, waves, B, lds_bytes, stream, P, "cone_dense_kernel");
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
  CAVE_LAUNCH(pack, waves, B, lds_bytes, stream, P, "cone_pack_kernel(count)");
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
  CAVE_LAUNCH(pack, waves, B, lds_bytes, stream, P, "cone_pack_kernel(fill)");
  return CAVE_OK;
}

int32_t cave_hip_packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz, int32_t all_pm1) {
  if (d <= 0 || max_rows < 0 || max_nnz < 0) return CAVE_E_INVALID;
  int32_t s = all_pm1 == 3 ? packed_lds_bytes(d, max_rows, max_nnz, true, false, true)
                           : packed_lds_bytes(d, max_rows, max_nnz, all_pm1 != 0, all_pm1 != 2);
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
  CAVE_LAUNCH(packed, waves, B, lds_bytes, stream, P, "cone_packed_kernel");
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

int64_t cave_hip_packed_large_rb_bytes(int64_t max_rows) { return (int64_t)rb_cache_bytes(max_rows); }

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

#define CAVE_LAUNCH_LARGE(FN, B, SLOTS, LDS, STREAM, PARAMS, WS, WHAT)                                          \
  do {                                                                                                         \
    const unsigned grid_ = (unsigned)((B) < (int64_t)(SLOTS) ? (B) : (int64_t)(SLOTS));                        \
    hipError_t e_ = FN(grid_, (uint32_t)(LDS), (hipStream_t)(STREAM), PARAMS, WS);                             \
    if (e_ != hipSuccess) return fail(CAVE_E_LAUNCH, "launch " WHAT, e_);                                      \
  } while (0)

int32_t cave_hip_cone_dense_large(const float* ctrs, const float* pred, int64_t B, int64_t m_max, int64_t d,
                                  int32_t mode, float sign, float inner_ratio, int32_t max_iter, int64_t nnz_cap,
                                  int32_t lds_bytes, void* workspace, int64_t slice_bytes, int32_t n_slots, float* proj,
                                  float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                                  int32_t* iters, void* stream) {
  if (B < 0 || m_max < 0 || m_max > 65535 || d <= 0 || d > 65535 || m_max * d >= (int64_t)1 << 32)
    return fail(CAVE_E_INVALID, "cone_dense_large: bad shape (need m_max, d <= 65535, m_max*d < 2^32)");
  if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_INNER_IPM) return fail(CAVE_E_INVALID, "cone_dense_large: bad mode");
  if (B == 0) return CAVE_OK;
  if (!ctrs && m_max > 0) return fail(CAVE_E_INVALID, "cone_dense_large: ctrs is null");
  if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_dense_large: pred is null");
  if (nnz_cap <= 0 || nnz_cap >= (int64_t)1 << 31) return fail(CAVE_E_INVALID, "cone_dense_large: bad nnz_cap");
  int32_t rc = check_large("cone_dense_large", workspace, slice_bytes, n_slots, lds_bytes);
  if (rc != CAVE_OK) return rc;
  DenseParams P;
  P.ctrs = ctrs; P.pred = pred; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d; P.mode = mode;
  P.sign = sign; P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100);
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  LargeWs W{(unsigned char*)workspace, (uint64_t)slice_bytes};
  CAVE_LAUNCH_LARGE(launch_dense_large, B, n_slots, lds_bytes, stream, P, W, "cone_dense_large_kernel");
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
  CAVE_LAUNCH_LARGE(launch_pack_large, B, n_slots, lds, stream, P, W, "cone_pack_large_kernel");
  return CAVE_OK;
}

int32_t cave_hip_cone_packed_large(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                                   int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                                   int32_t waves, void* workspace, int64_t slice_bytes, int32_t n_slots, float* proj,
                                   float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                                   int32_t* iters, void* stream) {
  if (!store || B < 0) return fail(CAVE_E_INVALID, "cone_packed_large: null store / bad B");
  if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_INNER_IPM) return fail(CAVE_E_INVALID, "cone_packed_large: bad mode");
  if (B == 0) return CAVE_OK;
  if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_packed_large: pred is null");
  int32_t rc = check_large("cone_packed_large", workspace, slice_bytes, n_slots, lds_bytes);
  if (rc != CAVE_OK) return rc;
  PackedParams P;
  P.store = *store; P.ids = ids; P.pred = pred; P.B = B; P.mode = mode; P.sign = sign;
  P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100); P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  LargeWs W{(unsigned char*)workspace, (uint64_t)slice_bytes};
  // workgroup shape (waves = 0): 4 waves (two workgroups per CU) unless the batch needs more than two workgroups
  // per CU and four fit the LDS; then 2 waves.
  if (waves != 0 && waves != 1 && waves != 2 && waves != 4)
    return fail(CAVE_E_INVALID, "cone_packed_large: waves must be 0, 1, 2 or 4");
  if (waves == 0) {
    const int64_t grid = B < n_slots ? B : n_slots;
    waves = (grid > 2 * (int64_t)device_cus() && (int64_t)lds_bytes * 4 <= (int64_t)kMaxLds) ? 2 : 4;
  }
  if (waves == 1) CAVE_LAUNCH_LARGE(launch_packed_large_w1, B, n_slots, lds_bytes, stream, P, W, "cone_packed_large_kernel<1>");
  else if (waves == 2) CAVE_LAUNCH_LARGE(launch_packed_large_w2, B, n_slots, lds_bytes, stream, P, W, "cone_packed_large_kernel<2>");
  else CAVE_LAUNCH_LARGE(launch_packed_large_w4, B, n_slots, lds_bytes, stream, P, W, "cone_packed_large_kernel<4>");
  return CAVE_OK;
}

// ------------------------------------------------------------------ fused step

static int32_t step_limits(int64_t m_max, int64_t d, int32_t& cap, int32_t& lds) {
  if (m_max < 0 || d <= 0 || d > kLiteMaxD || m_max > 32767) return CAVE_E_INVALID;
  cap = 0;
  uint64_t pack_lds = 0;
  if (m_max > 0) {
    // non-zeros kept per instance: structured cones carry <= d unit entries + a few sparse rows (cone_instance.h
    // default_limits); the arena of the two-wave pack half: no dump slots behind the scan output, no prediction
    int64_t c = 4 * (m_max + d) + 128;
    if (c > m_max * d) c = m_max * d;
    if (c < 64) c = 64;
    cap = (int32_t)c;
    pack_lds = arena_bytes_dense(m_max, d, c, 64, 32, c * 6 / 10, 0, true, false) - align8u(4 * d);
  }
  const uint32_t solve_lds = step_solve_lds_bytes(d);
  uint32_t need = pack_lds > solve_lds ? (uint32_t)pack_lds : solve_lds;
  need = (need + 255u) & ~255u;
  // four solve blocks + two pack blocks per compute unit: the fused form only pays when six workgroups fit
  // (a launch without a pack half -- m_max = 0: the lite slots of a device-resident store -- needs four)
  if ((uint64_t)need * (m_max > 0 ? 6u : 4u) > kMaxLds) return CAVE_E_INVALID;
  lds = (int32_t)need;
  return CAVE_OK;
}

int32_t cave_hip_step_lds_bytes(int64_t m_max, int64_t d) {
  int32_t cap = 0, lds = 0;
  const int32_t rc = step_limits(m_max, d, cap, lds);
  return rc == CAVE_OK ? lds : rc;
}

static bool lite_store_ok(const cave_lite_store* s, int64_t need, int64_t d) {
  return s && s->n >= need && s->d == d && s->hdr && s->usign && s->avg && s->rowptr && s->ell && s->csr16 && s->rl &&
         (((uintptr_t)s->ell | (uintptr_t)s->csr16) & 15u) == 0 && ((4 * d) & 3) == 0;
}

int32_t cave_hip_cone_step(const cave_lite_store* solve, const int64_t* ids, const float* pred, int64_t B, int32_t mode, float sign,
                           float inner_ratio, int32_t max_iter, int32_t flags, float* proj, float* rnorm, float* target, float* loss,
                           float* grad, int32_t* status, int32_t* iters, const float* next_ctrs, int64_t B_next,
                           int64_t m_max, int64_t d, const cave_lite_store* next, int32_t* pack_status,
                           uint32_t* cu_tickets, void* stream) {
  if (B < 0 || B_next < 0 || B + B_next >= (int64_t)1 << 31) return fail(CAVE_E_INVALID, "cone_step: bad batch sizes");
  if (B == 0 && B_next == 0) return CAVE_OK;
  if (d <= 0 || d > kLiteMaxD) return fail(CAVE_E_INVALID, "cone_step: need 0 < d <= 256");
  if (!cu_tickets) return fail(CAVE_E_INVALID, "cone_step: cu_tickets is null");
  StepParams P;
  memset(&P, 0, sizeof(P));
  if (B > 0) {
    if (mode < CAVE_MODE_PROJECT || mode > CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_step: bad mode (PROJECT .. AVG)");
    if (!pred && mode != CAVE_MODE_AVG) return fail(CAVE_E_INVALID, "cone_step: pred is null");
    if (!lite_store_ok(solve, ids ? 1 : B, d)) return fail(CAVE_E_INVALID, "cone_step: bad solve store (size, d, null or unaligned array)");
    P.S.store = *solve; P.S.ids = ids; P.S.pred = pred; P.S.B = B; P.S.mode = mode; P.S.sign = sign; P.S.inner_ratio = inner_ratio;
    P.S.max_iter = max_iter > 0 ? max_iter : 100;
    P.S.flags = flags;
    P.S.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  }
  int32_t cap = 0, lds = 0;
  if (step_limits(B_next > 0 ? m_max : 0, d, cap, lds) != CAVE_OK)
    return fail(CAVE_E_INVALID, "cone_step: shape does not qualify (cave_hip_step_lds_bytes)");
  if (B_next > 0) {
    if (m_max <= 0 || m_max * d >= (int64_t)1 << 32) return fail(CAVE_E_INVALID, "cone_step: bad m_max");
    if (!next_ctrs) return fail(CAVE_E_INVALID, "cone_step: next_ctrs is null");
    if (!lite_store_ok(next, B_next, d)) return fail(CAVE_E_INVALID, "cone_step: bad next store (size, d, null or unaligned array)");
    if (B > 0 && next->hdr == solve->hdr) return fail(CAVE_E_INVALID, "cone_step: solve and next must be different stores");
    P.Q.ctrs = next_ctrs; P.Q.B = B_next; P.Q.m = (int32_t)m_max; P.Q.d = (int32_t)d; P.Q.nnz_cap = (uint32_t)cap;
    P.Q.store = *next; P.Q.status = pack_status;
  }
  P.lds_bytes = (uint32_t)lds;
  P.tickets = cu_tickets;
  hipError_t e = launch_step((unsigned)(B + B_next), (uint32_t)lds, (hipStream_t)stream, P);
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "launch cone_step_kernel", e);
  return CAVE_OK;
}

int32_t cave_hip_lite_from_packed(const cave_cone_store* src, const cave_lite_store* dst, int32_t* status, void* stream) {
  if (!src || !dst) return fail(CAVE_E_INVALID, "lite_from_packed: null store");
  if (src->n == 0) return CAVE_OK;
  if (src->n < 0 || src->n >= (int64_t)1 << 31 || src->d <= 0 || src->d > kLiteMaxD)
    return fail(CAVE_E_INVALID, "lite_from_packed: need 0 < d <= 256, n < 2^31");
  if (!lite_store_ok(dst, src->n, src->d)) return fail(CAVE_E_INVALID, "lite_from_packed: bad lite store (size, d, null or unaligned array)");
  const int64_t d = src->d;
  const uint64_t lds = 256 + 64 + align8u(4 * d) + align8u(d) + align8u(4 * (d + 1)) + align8u(4 * (kLiteMaxRows + 1)) + 64 +
                       2 * align8u(2 * 64 * kLiteMaxChunk) + lite_lds_bytes((int)d, 64u * kLiteMaxChunk) + 64;
  LiteFromPackedParams P;
  P.src = *src; P.dst = *dst; P.n = src->n; P.lds_bytes = (uint32_t)lds; P.status = status;
  hipError_t e = launch_lite_from_packed((unsigned)src->n, (uint32_t)lds, (hipStream_t)stream, P);
  if (e != hipSuccess) return fail(CAVE_E_LAUNCH, "launch lite_from_packed_kernel", e);
  return CAVE_OK;
}

}  // extern "C"
