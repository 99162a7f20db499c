// emul_abi.cpp — serial gcc build of the per-instance code (TEST INFRASTRUCTURE ONLY).
// Exports cave_emul_* with the signatures of include/cave_hip.h minus the stream;
// all pointers are HOST pointers.  Used by tests/ to exercise the shared
// algorithm code under sanitizers without a GPU; never loaded by cave_amd.
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CAVE_EMUL_COUNTERS 1
#include "ctx_serial.h"
#include "../../cave_amd/csrc/cone_core.h"
#include "../../cave_amd/csrc/cone_instance.h"

using namespace cave;

extern "C" {

// how many instances took which path since the last call ([0] dense LDL^T of the large-cone path, ...); resets
void cave_emul_path_counters(long* out) {
  for (int i = 0; i < 8; ++i) { out[i] = emul_counters()[i]; emul_counters()[i] = 0; }
}

int32_t cave_emul_default_limits(int64_t m_max, int64_t d, int32_t* nnz_cap, int32_t* lds_bytes) {
  return default_limits(m_max, d, nnz_cap, lds_bytes);
}

int32_t cave_emul_cone_dense(const float* ctrs, const float* pred, int64_t B, int64_t m_max, int64_t d, int32_t mode,
                             float sign, float inner_ratio, int32_t max_iter, int32_t nnz_cap, int32_t lds_bytes,
                             float* proj, float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                             int32_t* iters) {
  // explicit limits are taken as given, even beyond the 160 KiB a real workgroup has: this lets
  // the CPU tier validate the algorithm at sizes (TSP-50) the round-1 kernels cannot hold in LDS yet
  if (!(nnz_cap > 0 && lds_bytes > 0) && !resolve_limits(m_max, d, nnz_cap, lds_bytes)) return CAVE_E_INVALID;
  DenseParams P;
  P.ctrs = ctrs; P.pred = pred; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d; P.mode = mode;
  P.sign = sign; P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : 100;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  std::vector<unsigned char> smem((size_t)lds_bytes);  // exact size: ASan catches arena overruns
  SerialCtx c;
  for (int64_t b = 0; b < B; ++b) run_dense_instance(c, smem.data(), P, b);
  return CAVE_OK;
}

int32_t cave_emul_pack_count(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int32_t nnz_cap,
                             int32_t lds_bytes, int32_t* n_rows, int32_t* n_nnz, int32_t* status) {
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes)) return CAVE_E_INVALID;
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.n_rows = n_rows; P.n_nnz = n_nnz; P.status = status; P.fill = 0;
  std::vector<unsigned char> smem((size_t)lds_bytes);
  SerialCtx c;
  for (int64_t b = 0; b < B; ++b) run_pack_instance(c, smem.data(), P, b);
  return CAVE_OK;
}

int32_t cave_emul_pack_fill(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int32_t nnz_cap,
                            int32_t lds_bytes, const cave_cone_store* store, int64_t slot0, int32_t* status) {
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes)) return CAVE_E_INVALID;
  if (!store || store->d != d || slot0 < 0 || slot0 + B > store->n) return CAVE_E_INVALID;
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.status = status; P.store = *store; P.slot0 = slot0; P.fill = 1;
  std::vector<unsigned char> smem((size_t)lds_bytes);
  SerialCtx c;
  for (int64_t b = 0; b < B; ++b) run_pack_instance(c, smem.data(), P, b);
  return CAVE_OK;
}

int32_t cave_emul_packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz, int32_t all_pm1) {
  int32_t s = packed_lds_bytes(d, max_rows, max_nnz, all_pm1 != 0);
  return s < 0 ? CAVE_E_INVALID : s;
}

int32_t cave_emul_cone_packed(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                              int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                              float* proj, float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                              int32_t* iters) {
  if (!store || lds_bytes <= 0) return CAVE_E_INVALID;
  PackedParams P;
  P.store = *store; P.ids = ids; P.pred = pred; P.B = B; P.mode = mode; P.sign = sign;
  P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : 100; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  std::vector<unsigned char> smem((size_t)lds_bytes);
  SerialCtx c;
  for (int64_t b = 0; b < B; ++b) run_packed_instance(c, smem.data(), P, b);
  return CAVE_OK;
}

// band LDL^T solver alone (cone_band.h): Hb [p*(bw+1)] band form, x out
int32_t cave_emul_band_solve(const double* Hb, int32_t bw, const double* rhs, const uint8_t* act, int32_t p,
                             double reg_rel, double* x) {
  const size_t ld = (size_t)bw + 1;
  std::vector<double> win(ld * ld), fac((size_t)(p > 0 ? p : 1) * ld), z((size_t)(p > 0 ? p : 1));
  SerialCtx c;
  int ch = band_chunk_rows<SerialCtx>((int)ld);
  if (ch > 3) ch = 3;  // small chunks: exercise the chunk hand-over often
  std::vector<double> stg(2 * (size_t)ch * ld);
  solve_spd_band<SerialCtx, false>(c, Hb, bw, rhs, act, p, reg_rel, win.data(), fac.data(), z.data(), x, stg.data(), ch);
  return CAVE_OK;
}

// ---- large-cone path: one serial "workgroup", arena = heap slice, hot arena = lds_bytes of heap

int64_t cave_emul_large_slice_bytes(int64_t m_max, int64_t d, int64_t nnz_cap, int64_t band_entries) {
  return (int64_t)large_slice_bytes(m_max, d, nnz_cap, band_entries);
}
int64_t cave_emul_packed_large_slice_bytes(int64_t d, int64_t max_rows, int64_t band_entries) {
  return (int64_t)packed_large_slice_bytes(d, max_rows, band_entries);
}

int32_t cave_emul_cone_dense_large(const float* ctrs, const float* pred, int64_t B, int64_t m_max, int64_t d,
                                   int32_t mode, float sign, float inner_ratio, int32_t max_iter, int64_t nnz_cap,
                                   int32_t lds_bytes, int64_t slice_bytes, float* proj, float* rnorm, float* target,
                                   float* loss, float* grad, int32_t* status, int32_t* iters) {
  if (nnz_cap <= 0 || lds_bytes <= 0 || slice_bytes <= 0 || slice_bytes >= ((int64_t)1 << 32)) return CAVE_E_INVALID;
  DenseParams P;
  P.ctrs = ctrs; P.pred = pred; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d; P.mode = mode;
  P.sign = sign; P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100);
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  std::vector<unsigned char> smem((size_t)lds_bytes), ws((size_t)slice_bytes);
  SerialCtx c;
  for (int64_t b = 0; b < B; ++b) run_dense_instance<SerialCtx, true>(c, smem.data(), P, b, ws.data(), (uint32_t)slice_bytes);
  return CAVE_OK;
}

int32_t cave_emul_pack_large(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int64_t nnz_cap,
                             int64_t slice_bytes, int32_t* n_rows, int32_t* n_nnz, const cave_cone_store* store,
                             int64_t slot0, int32_t* status) {
  if (nnz_cap <= 0 || slice_bytes <= 0 || slice_bytes >= ((int64_t)1 << 32)) return CAVE_E_INVALID;
  if (store && (store->d != d || slot0 < 0 || slot0 + B > store->n)) return CAVE_E_INVALID;
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = 1024;
  P.n_rows = n_rows; P.n_nnz = n_nnz; P.status = status;
  if (store) { P.store = *store; P.slot0 = slot0; P.fill = 1; }
  std::vector<unsigned char> smem(1024), ws((size_t)slice_bytes);
  SerialCtx c;
  for (int64_t b = 0; b < B; ++b) run_pack_instance<SerialCtx, true>(c, smem.data(), P, b, ws.data(), (uint32_t)slice_bytes);
  return CAVE_OK;
}

int32_t cave_emul_cone_packed_large(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                                    int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                                    int64_t slice_bytes, float* proj, float* rnorm, float* target, float* loss,
                                    float* grad, int32_t* status, int32_t* iters) {
  if (!store || lds_bytes <= 0 || slice_bytes <= 0 || slice_bytes >= ((int64_t)1 << 32)) return CAVE_E_INVALID;
  PackedParams P;
  P.store = *store; P.ids = ids; P.pred = pred; P.B = B; P.mode = mode; P.sign = sign;
  P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100); P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  std::vector<unsigned char> smem((size_t)lds_bytes), ws((size_t)slice_bytes);
  SerialCtx c;
  for (int64_t b = 0; b < B; ++b) run_packed_large_instance<SerialCtx>(c, smem.data(), P, b, ws.data(), (uint32_t)slice_bytes);
  return CAVE_OK;
}

}  // extern "C"
