// simt_abi.cpp — SIMT emulation build of the GPU code paths (TEST INFRASTRUCTURE ONLY).
//
// Compiles cave_amd/csrc exactly as the HIP build sees it -- wave contexts, DPP / readlane / ballot primitives,
// the one-wave lite solver, the one-/two-wave band elimination, the blocked dense LDL^T -- with g++ against the
// shim in tests/emul/simt/hip/hip_runtime.h: every lane of a workgroup is a fiber, every cross-lane primitive and
// barrier a rendezvous.  Runs under -fsanitize=address,undefined in the CPU test tier.  Exports cave_simt_* with the
// signatures of include/cave_hip.h (host pointers, no stream, + a schedule seed: 0 = round robin, else the lanes
// between two rendezvous run in a seeded random order).  Never loaded by cave_amd.
#define CAVE_SIMT_EMUL 1
#define CAVE_EMUL_COUNTERS 1
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/cave_hip.h"
#include "../../cave_amd/csrc/cone_common.h"
#include "../../cave_amd/csrc/cone_core.h"
#include "../../cave_amd/csrc/ctx_wave.h"
#include "../../cave_amd/csrc/ctx_block.h"
#include "../../cave_amd/csrc/cone_instance.h"

using namespace cave;

namespace {

using Ctx1 = WaveCtx;
using Ctx2 = BlockCtx<2>;
using Ctx4 = BlockCtx<4>;
using CtxW = BlockCtx<4, true>;
using CtxL = BlockCtx<4, true>;
using CtxL2 = BlockCtx<2, true>;

struct Lds {  // exact-size, 16-byte aligned heap block standing in for the workgroup's LDS: ASan sees overruns
  std::vector<unsigned char> raw;
  unsigned char* p;
  explicit Lds(size_t n) : raw(n + 16) {
    uintptr_t a = (uintptr_t)raw.data();
    size_t off = (16 - (a & 15)) & 15;
    p = raw.data() + off;
    raw.resize(off + n);  // no slack behind the arena
    p = raw.data() + off;
  }
};

template <class C, class F>
void launch(int64_t B, unsigned grid, uint64_t seed, F&& body) {  // one workgroup per instance
  for (int64_t b = 0; b < B; ++b) {
    simt::run_block(C::NT, (unsigned)b, grid, [&]() { body(b); }, seed ? seed + (uint64_t)b : 0);
  }
}

template <class C>
int32_t packed_impl(const PackedParams& P, uint64_t seed) {
  Lds lds(P.lds_bytes);
  launch<C>(P.B, (unsigned)P.B, seed, [&](int64_t b) {
    C c;
    c.init(lds.p);
    run_packed_instance(c, lds.p, P, b);
  });
  return CAVE_OK;
}

template <class C>
int32_t dense_impl(const DenseParams& P, uint64_t seed) {
  Lds lds(P.lds_bytes);
  launch<C>(P.B, (unsigned)P.B, seed, [&](int64_t b) {
    C c;
    c.init(lds.p);
    run_dense_instance(c, lds.p, P, b);
  });
  return CAVE_OK;
}

template <class C>
int32_t pack_impl(const PackParams& P, uint64_t seed) {
  Lds lds(P.lds_bytes);
  launch<C>(P.B, (unsigned)P.B, seed, [&](int64_t b) {
    C c;
    c.init(lds.p);
    run_pack_instance(c, lds.p, P, b);
  });
  return CAVE_OK;
}

template <class C>
int32_t packed_large_impl(const PackedParams& P, int64_t slice_bytes, uint64_t seed) {
  Lds lds(P.lds_bytes);
  std::vector<unsigned char> ws((size_t)slice_bytes + 16);
  unsigned char* wsp = ws.data() + ((16 - ((uintptr_t)ws.data() & 15)) & 15);
  launch<C>(P.B, (unsigned)P.B, seed, [&](int64_t b) {
    C c;
    c.init(lds.p);
    run_packed_large_instance<C>(c, lds.p, P, b, wsp, (uint32_t)slice_bytes);
  });
  return CAVE_OK;
}

}  // namespace

extern "C" {

void cave_simt_path_counters(long* out) {
  for (int i = 0; i < 8; ++i) { out[i] = emul_counters()[i]; emul_counters()[i] = 0; }
}
void cave_simt_stats(unsigned long* out) {  // context switches, rendezvous since the start
  out[0] = simt::S().switches;
  out[1] = simt::S().rendezvous;
}

int32_t cave_simt_packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz, int32_t all_pm1) {
  int32_t s = all_pm1 == 3 ? packed_lds_bytes(d, max_rows, max_nnz, true, false, true)   // the diet layout
                           : packed_lds_bytes(d, max_rows, max_nnz, all_pm1 != 0, all_pm1 != 2);
  return s < 0 ? CAVE_E_INVALID : s;
}
int32_t cave_simt_packed_large_lds_bytes(int32_t max_rows, int32_t max_bw) {
  return (int32_t)packed_large_lds_bytes(max_rows, max_bw);
}
int64_t cave_simt_packed_large_slice_bytes(int64_t d, int64_t max_rows, int64_t band_entries) {
  return (int64_t)packed_large_slice_bytes(d, max_rows, band_entries);
}

int32_t cave_simt_cone_dense(const float* ctrs, const float* pred, int64_t B, int64_t m_max, int64_t d, int32_t mode,
                             float sign, float inner_ratio, int32_t max_iter, int32_t nnz_cap, int32_t lds_bytes,
                             int32_t waves, uint64_t seed, float* proj, float* rnorm, float* target, float* loss,
                             float* grad, int32_t* status, int32_t* iters) {
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes, waves <= 2 && B <= 2048)) return CAVE_E_INVALID;
  DenseParams P;
  P.ctrs = ctrs; P.pred = pred; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d; P.mode = mode;
  P.sign = sign; P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100);
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  switch (waves) {
    case 1: return dense_impl<Ctx1>(P, seed);
    case 2: return dense_impl<Ctx2>(P, seed);
    case 4: return dense_impl<Ctx4>(P, seed);
    case 8: return dense_impl<CtxW>(P, seed);
  }
  return CAVE_E_INVALID;
}

int32_t cave_simt_pack_fill(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int32_t nnz_cap, int32_t lds_bytes,
                            int32_t waves, uint64_t seed, const cave_cone_store* store, int64_t slot0, int32_t* n_rows,
                            int32_t* n_nnz, int32_t* status) {
  if (!resolve_limits(m_max, d, nnz_cap, lds_bytes, false)) return CAVE_E_INVALID;
  PackParams P;
  memset(&P, 0, sizeof(P));
  P.ctrs = ctrs; P.B = B; P.m = (int32_t)m_max; P.d = (int32_t)d;
  P.nnz_cap = (uint32_t)nnz_cap; P.lds_bytes = (uint32_t)lds_bytes;
  P.n_rows = n_rows; P.n_nnz = n_nnz; P.status = status;
  if (store) { P.store = *store; P.slot0 = slot0; P.fill = 1; }
  switch (waves) {
    case 1: return pack_impl<Ctx1>(P, seed);
    case 2: return pack_impl<Ctx2>(P, seed);
    case 4: return pack_impl<Ctx4>(P, seed);
    case 8: return pack_impl<CtxW>(P, seed);
  }
  return CAVE_E_INVALID;
}

int32_t cave_simt_cone_packed(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                              int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                              int32_t waves, uint64_t seed, float* proj, float* rnorm, float* target, float* loss,
                              float* grad, int32_t* status, int32_t* iters) {
  if (!store || lds_bytes <= 0) return CAVE_E_INVALID;
  PackedParams P;
  P.store = *store; P.ids = ids; P.pred = pred; P.B = B; P.mode = mode; P.sign = sign;
  P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100);
  P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  switch (waves) {
    case 1: return packed_impl<Ctx1>(P, seed);
    case 2: return packed_impl<Ctx2>(P, seed);
    case 4: return packed_impl<Ctx4>(P, seed);
    case 8: return packed_impl<CtxW>(P, seed);
  }
  return CAVE_E_INVALID;
}

int32_t cave_simt_cone_packed_large(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                                    int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                                    int32_t waves, int64_t slice_bytes, uint64_t seed, float* proj, float* rnorm,
                                    float* target, float* loss, float* grad, int32_t* status, int32_t* iters) {
  if (!store || lds_bytes <= 0 || slice_bytes <= 0 || slice_bytes >= ((int64_t)1 << 32)) return CAVE_E_INVALID;
  PackedParams P;
  P.store = *store; P.ids = ids; P.pred = pred; P.B = B; P.mode = mode; P.sign = sign;
  P.inner_ratio = inner_ratio; P.max_iter = max_iter > 0 ? max_iter : (mode == CAVE_MODE_INNER_IPM ? 3 : 100);
  P.lds_bytes = (uint32_t)lds_bytes;
  P.o = OutPtrs{proj, rnorm, target, loss, grad, status, iters};
  switch (waves) {
    case 1: return packed_large_impl<Ctx1>(P, slice_bytes, seed);
    case 2: return packed_large_impl<CtxL2>(P, slice_bytes, seed);
    case 4: return packed_large_impl<CtxL>(P, slice_bytes, seed);
  }
  return CAVE_E_INVALID;
}

}  // extern "C"
