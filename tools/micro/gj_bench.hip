// Micro-benchmark: cycles per call of the register Gauss-Jordan solve variants (one wave, p x p SPD system).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../../cave_amd/csrc/wave_prims.h"
using namespace cave;

// round-1 implementation, kept here as the timing / accuracy reference
template <int PM>
__device__ __forceinline__ void gj_ref_regs(int lane, const double* H, int ldh, const double* rhs,
                                              const uint8_t* act, int p, double reg_rel, double* dv) {
  const bool live = lane < p;
  const bool my_act = live && act[lane] != 0;
  double diag0 = (live && !my_act) ? H[lane * ldh + lane] : 0.0;
  const double maxdiag = wave_max_f64(diag0);
  const double reg = reg_rel * maxdiag;
  double h[PM];
#pragma unroll
  for (int j = 0; j < PM; ++j) {
    double v = 0.0;
    if (j < p) {
      if (live && !my_act) v = H[lane * ldh + j];
      if (j == lane) v = my_act ? 1.0 : v + reg;
    }
    h[j] = v;
  }
  double b = live ? rhs[lane] : 0.0;
  double diag = 1.0;
  bool dead = !live;
#pragma unroll
  for (int k = 0; k < PM; ++k) {
    if (k < p) {
      const double piv = readlane_f64(h[k], k);
      const double bk = readlane_f64(b, k);
      // reciprocal by v_rcp_f64 + two Newton steps (full double accuracy, a third of the IEEE divide);
      // a non-positive / NaN pivot (numerically dependent row) gives inv = 0: the row is skipped, x_k = 0
      const bool ok = piv > 1e-300;
      double inv = __builtin_amdgcn_rcp(ok ? piv : 1.0);
      inv = fma(fma(-piv, inv, 1.0), inv, inv);
      inv = fma(fma(-piv, inv, 1.0), inv, inv);
      inv = ok ? inv : 0.0;
      if (lane == k) { diag = ok ? piv : 1.0; dead = dead || !ok; }
      const double fac = (lane == k) ? 0.0 : h[k] * inv;
#pragma unroll
      for (int j = k + 1; j < PM; ++j) h[j] -= fac * readlane_f64(h[j], k);  // columns >= p hold zeros
      b -= fac * bk;
    }
  }
  if (live) dv[lane] = dead ? 0.0 : b / diag;
}

// size-specialised dispatch; PLIM bounds the register footprint (2*PM VGPRs for the row)
template <int PLIM>
__device__ __forceinline__ void gj_ref(int lane, const double* H, int ldh, const double* g, const uint8_t* act, int p,
                                         double reg_rel, double* dv) {
  if (p <= 8) gj_ref_regs<8>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 16) gj_ref_regs<16>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 20) gj_ref_regs<20>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 24) gj_ref_regs<24>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 28) gj_ref_regs<28>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if (p <= 32) gj_ref_regs<32>(lane, H, ldh, g, act, p, reg_rel, dv);
  else if constexpr (PLIM > 32) {
    if (p <= 40) gj_ref_regs<40>(lane, H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 48) gj_ref_regs<48>(lane, H, ldh, g, act, p, reg_rel, dv);
    else if (p <= 56) gj_ref_regs<56>(lane, H, ldh, g, act, p, reg_rel, dv);
    else gj_ref_regs<64>(lane, H, ldh, g, act, p, reg_rel, dv);
  }
}



template <int VAR>
__global__ __launch_bounds__(64) void k_gj(const double* Hg, const double* rhs, int p, int ldh, int reps, double* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* H = (double*)smem;
  double* g = H + p * ldh;
  double* dv = g + 64;
  uint8_t* act = (uint8_t*)(dv + 64);
  const int lane = threadIdx.x;
  for (int i = lane; i < p * ldh; i += 64) H[i] = Hg[blockIdx.x * p * ldh + i];
  if (lane < p) { g[lane] = rhs[blockIdx.x * p + lane]; act[lane] = 0; }
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (VAR == 0) gj_ref<32>(lane, H, ldh, g, act, p, 1e-12, dv);
    if (VAR == 1) gj_solve<32>(lane, H, ldh, g, act, p, 1e-12, dv);
    if (VAR == 2) gj_solve<32, true>(lane, H, ldh, g, act, p, 1e-12, dv);
    __syncthreads();
    if (lane < p) g[lane] += 1e-9 * dv[lane];  // dependency between repetitions
    __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane < p) out[blockIdx.x * p + lane] = dv[lane];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  int p = argc > 1 ? atoi(argv[1]) : 24;
  int reps = 50, nb = 1024, ldh = p | 1;
  std::vector<double> H((size_t)nb * p * ldh), b((size_t)nb * p);
  srand(1);
  for (int blk = 0; blk < nb; ++blk) {
    std::vector<double> M(p * 40);
    for (auto& x : M) x = (rand() % 3) - 1;
    for (int i = 0; i < p; ++i)
      for (int j = 0; j < p; ++j) {
        double s = (i == j) ? 0.5 : 0.0;
        for (int k = 0; k < 40; ++k) s += M[i * 40 + k] * M[j * 40 + k];
        H[(size_t)blk * p * ldh + i * ldh + j] = s;
      }
    for (int i = 0; i < p; ++i) b[(size_t)blk * p + i] = (rand() % 1000) / 500.0 - 1.0;
  }
  double *dH, *db, *dout; unsigned long long* dc;
  hipMalloc(&dH, H.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dout, b.size() * 8); hipMalloc(&dc, nb * 8);
  hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
  size_t lds = (size_t)(p * ldh + 128) * 8 + 64;
  std::vector<double> o0(b.size()), o1(b.size());
  std::vector<unsigned long long> c(nb);
  std::vector<double> o2(b.size());
  for (int var = 0; var < 3; ++var) {
    for (int rep = 0; rep < 2; ++rep) {
      if (var == 0) hipLaunchKernelGGL(k_gj<0>, dim3(nb), dim3(64), lds, 0, dH, db, p, ldh, reps, dout, dc);
      else if (var == 1) hipLaunchKernelGGL(k_gj<1>, dim3(nb), dim3(64), lds, 0, dH, db, p, ldh, reps, dout, dc);
      else hipLaunchKernelGGL(k_gj<2>, dim3(nb), dim3(64), lds, 0, dH, db, p, ldh, reps, dout, dc);
      hipDeviceSynchronize();
    }
    hipMemcpy((var == 2 ? o2 : var ? o1 : o0).data(), dout, b.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, nb * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : c) s += x;
    printf("variant %d  p=%d: %.0f cycles per solve (%.0f per pivot)\n", var, p, s / nb / reps, s / nb / reps / p);
  }
  double md = 0; for (size_t i = 0; i < o0.size(); ++i) md = fmax(md, fabs(o0[i] - o1[i]) / (1e-30 + fabs(o0[i])));
  // residual check of variant 1 on block 0
  double rmax = 0;
  for (int i = 0; i < p; ++i) { double s = -b[i]; for (int j = 0; j < p; ++j) s += H[i * ldh + j] * o1[j]; rmax = fmax(rmax, fabs(s)); }
  double md2 = 0; for (size_t i = 0; i < o0.size(); ++i) md2 = fmax(md2, fabs(o0[i] - o2[i]) / (1e-30 + fabs(o0[i])));
  printf("max rel diff v0 vs v1 %.2e, v0 vs v3 %.2e; residual v1 blk0 %.2e\n", md, md2, rmax);
  return 0;
}
