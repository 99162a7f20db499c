// ctx_serial.h — single-lane stand-in for WaveCtx (TEST INFRASTRUCTURE ONLY).
//
// Lets gcc (with -fsanitize=address,undefined) execute the exact control flow
// of cave_amd/csrc/cone_core.h + cone_instance.h on a machine without a GPU.
// Nothing in the cave_amd package loads this build.
#pragma once
#include <string.h>
#include "../../cave_amd/csrc/cone_common.h"

namespace cave {

struct SerialCtx {
  static constexpr int NT = 1;
  static constexpr int TEAM = 1;
  static constexpr uint32_t SCRATCH_BYTES = 0;
  static constexpr int KREG = 0;  // always the LDS (memory) form of the line search
  void reduce_sum2(double&, double&) const {}
  double team_reduce_sum(double v) const { return v; }
  static constexpr int PMAX = 64;
  static constexpr int NWAVES = 1;
  static constexpr int WL = 1;  // single lane: the back substitution takes its strided form
  int tid() const { return 0; }
  int wave_id() const { return 0; }
  int lane_id() const { return 0; }
  double wave_sum(double v) const { return v; }
  double wave_max(double v) const { return v; }
  double wave_shift_up(double) const { return 0.0; }
  void wave_fence() const {}
  void wave_fence_lds() const {}
  void sync_lds() const {}
  void sync() const {}
  double reduce_sum(double v) const { return v; }
  double reduce_max(double v) const { return v; }
  uint32_t reduce_add_u32(uint32_t v) const { return v; }
  void atomic_add_u32(uint32_t* p, uint32_t v) const { *p += v; }
  uint32_t atomic_inc_ret_u32(uint32_t* p) const { return (*p)++; }
  void atomic_or_u32(uint32_t* p, uint32_t v) const { *p |= v; }
  void atomic_add_f64(double* p, double v) const { *p += v; }
  void atomic_add_f64_lds(double* p, double v) const { *p += v; }
  void atomic_add_i64_lds(long long* p, long long v) const { *p += v; }
  void atomic_add_i64(long long* p, long long v) const { *p += v; }
  uint32_t exclusive_scan_u32(uint32_t* a, int n) const {
    uint32_t run = 0;
    for (int i = 0; i < n; ++i) { uint32_t v = a[i]; a[i] = run; run += v; }
    return run;
  }
  uint32_t compact_mask_u8(const uint8_t* f, int n, uint8_t mask, uint8_t val, uint32_t* out) const {
    uint32_t c = 0;
    for (int i = 0; i < n; ++i) if ((f[i] & mask) == val) out[c++] = (uint32_t)i;
    return c;
  }
  uint32_t compact_nonzero_u8(const uint8_t* f, int n, uint32_t* out) const {
    uint32_t c = 0;
    for (int i = 0; i < n; ++i) if (f[i]) out[c++] = (uint32_t)i;
    return c;
  }
  template <bool COND = false>
  uint32_t scan_dense(const float* A, uint32_t n, uint32_t* eflat, float* eval, uint32_t cap) const {
    uint32_t cur = 0;
    for (uint32_t f = 0; f < n; ++f) {
      float v = A[f];
      if (v != 0.0f) {
        if (cur < cap) { eflat[cur] = f; eval[cur] = v; }
        cur++;
      }
    }
    return cur;
  }
  // same Gauss-Jordan elimination order as WaveCtx::solve_spd_regs
  void solve_spd(const double* H, int ldh, const double* rhs, const uint8_t* act, int p, double reg_rel,
                 double* dv) const {
    double maxdiag = 0.0;
    for (int i = 0; i < p; ++i) if (!act[i]) maxdiag = fmax(maxdiag, H[i * ldh + i]);
    const double reg = reg_rel * maxdiag;
    if (p <= 0) return;
    double* h = new double[(size_t)p * p];
    double* b = new double[p];
    double* diag = new double[p];
    bool* dead = new bool[p];
    for (int i = 0; i < p; ++i) {
      for (int j = 0; j < p; ++j) {
        double v = 0.0;
        if (!act[i]) v = H[i * ldh + j];
        if (i == j) v = act[i] ? 1.0 : v + reg;
        h[i * p + j] = v;
      }
      b[i] = rhs[i];
      diag[i] = 1.0;
      dead[i] = false;
    }
    for (int k = 0; k < p; ++k) {
      double piv = h[k * p + k], bk = b[k];
      if (piv > 1e-300) {
        diag[k] = piv;
        for (int i = 0; i < p; ++i) {
          if (i == k) continue;
          double fac = h[i * p + k] / piv;
          for (int j = k + 1; j < p; ++j) h[i * p + j] -= fac * h[k * p + j];
          b[i] -= fac * bk;
        }
      } else dead[k] = true;
    }
    for (int i = 0; i < p; ++i) dv[i] = dead[i] ? 0.0 : b[i] / diag[i];
    delete[] h; delete[] b; delete[] diag; delete[] dead;
  }
};

}  // namespace cave
