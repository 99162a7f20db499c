// Micro-benchmark: cycles of the pieces of one Newton iteration on ONE wave, on real TSP-20 cones
// (dense input -> the kernels' own scan / build / lite_build, then each piece timed with s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../include/cave_hip.h"
#include "../../cave_amd/csrc/cone_common.h"
#include "../../cave_amd/csrc/cone_core.h"
#include "../../cave_amd/csrc/ctx_wave.h"
#include "../../cave_amd/csrc/ctx_block.h"
#include "../../cave_amd/csrc/cone_instance.h"
using namespace cave;

#define NSLOT 19
__global__ __launch_bounds__(64) void k(const float* ctrs, const float* pred, int m, int d, uint32_t cap, uint32_t lds,
                                        unsigned long long* out, int reps, unsigned long long* out2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WaveCtx c; c.init(smem);
  Arena ar; ar.init(smem, lds);
  ConeBuild cb;
  const int b = blockIdx.x;
  int32_t st = scan_and_build<WaveCtx, false>(c, ar, cb, ctrs + (size_t)b * m * d, m, d, cap);
  if (st != ST_OK) { if (threadIdx.x == 0) out[b * NSLOT] = 999999999ull; return; }
  float* y = ar.get<float>(d);
  for (int kx = c.tid(); kx < d; kx += 64) y[kx] = -pred[(size_t)b * d + kx];
  ar.release_top();
  SolveView v = view_of(cb);
  const int p = v.p;
  double* res = ar.get<double>(d + 1); double* rc = ar.get<double>(d + 1); double* q = ar.get<double>(d + 1);
  double* theta = ar.get<double>(40); double* g = ar.get<double>(40); double* dv = ar.get<double>(40);
  double* H = ar.get<double>(33 * 33 + 8); uint8_t* act = ar.get<uint8_t>(40); float* wold = ar.get<float>(d);
  LiteCone L;
  c.sync();
  unsigned long long t[NSLOT + 1];
  t[0] = __builtin_amdgcn_s_memtime();
  bool ok = lite_build(c, ar, v, L);
  t[NSLOT] = __builtin_amdgcn_s_memtime();
  if (!ok) { if (threadIdx.x == 0) out[b * NSLOT] = 888888888ull; return; }
  SoloCtx<32, 4> sc; sc.lane = threadIdx.x; sc.lite = L;
  for (int kx = c.tid(); kx <= d; kx += 64) { res[kx] = kx < d ? y[kx] : 0.0; rc[kx] = kx < d ? clip_unit(y[kx], v.usign[kx]) : 0.0; }
  for (int i = c.tid(); i < 40; i += 64) { theta[i] = i < 32 ? 0.01 * i : 0.0; dv[i] = i < 32 ? 0.02 * i - 0.1 : 0.0; act[i] = 0; }
  for (int i = c.tid(); i < 33 * 33; i += 64) H[i] = (i % 34 == 0) ? 20.0 : 0.25;
  c.sync();
  t[1] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) lite_gradient(sc, L, p, rc, g);
  t[2] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) lite_gather(sc, L, d, y, theta, -1.0, res);
  t[3] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) { double f = refresh_clipped(sc, v, res, rc); if (f < -1.0) g[0] = f; }
  t[4] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) { sc.solve_spd(H, 33, g, act, p, 1e-12, dv); sc.sync(); }
  t[5] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) gradient<SoloCtx<32, 4>, true>(sc, v, rc, g);   // general form, for comparison
  t[6] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) gather_mt<SoloCtx<32, 4>, true>(sc, v, y, theta, -1.0, res);
  t[7] = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) { double a = sc.reduce_sum(g[threadIdx.x & 31]); double bb = sc.reduce_max(dv[threadIdx.x & 31]); if (a + bb == 1.2345) g[1] = a; }
  t[8] = __builtin_amdgcn_s_memtime();
  SolveWork w; w.H = H; w.ldh = 33; w.wold = wold; w.warm = nullptr;
  unsigned long long th0 = 0, th1 = 0;
  for (int r = 0; r < reps; ++r) {
    for (int kx = c.tid(); kx < d; kx += 64) wold[kx] = 0.f;   // every coordinate with weight > 0 changes
    sc.sync();
    unsigned long long a0 = __builtin_amdgcn_s_memtime();
    lite_hessian(sc, L, v, w, res, 0.05, 20.0);
    sc.sync();
    unsigned long long a1 = __builtin_amdgcn_s_memtime();
    lite_hessian(sc, L, v, w, res, 0.05, 20.0);                 // nothing changes
    sc.sync();
    unsigned long long a2 = __builtin_amdgcn_s_memtime();
    th0 += a1 - a0; th1 += a2 - a1;
  }
  // the same pieces interleaved (each call follows OTHER code, as inside the Newton loop)
  unsigned long long ti0 = 0, ti1 = 0;
  for (int r = 0; r < reps; ++r) {
    unsigned long long a0 = __builtin_amdgcn_s_memtime();
    lite_gradient(sc, L, p, rc, g);
    unsigned long long a1 = __builtin_amdgcn_s_memtime();
    sc.solve_spd(H, 33, g, act, p, 1e-12, dv); sc.sync();
    for (int kx = c.tid(); kx < d; kx += 64) wold[kx] = 0.f;
    sc.sync();
    lite_hessian(sc, L, v, w, res, 0.05, 20.0);
    sc.sync();
    unsigned long long a2 = __builtin_amdgcn_s_memtime();
    lite_gather(sc, L, d, y, theta, -1.0, res);
    unsigned long long a3 = __builtin_amdgcn_s_memtime();
    ti0 += a1 - a0; ti1 += a3 - a2;
  }
  // the whole lite solve, as the kernels run it
  double* ttry = ar.get<double>(40); double* told = ar.get<double>(40); double* g2 = ar.get<double>(40); double* step = ar.get<double>(40);
  w.y = y; w.res = res; w.q = q; w.rc = rc; w.theta = theta; w.ttry = ttry; w.told = told; w.g = g; w.dv = dv; w.g2 = g2; w.step = step;
  w.act = act; w.band_hot = false; w.bw = 0;
  if (threadIdx.x == 0) { rc[d] = 0.0; theta[32] = 0.0; dv[32] = 0.0; }
  sc.sync();
#ifdef CAVE_STAMPS
  unsigned long long stf[32];
  for (int i = 0; i < 32; ++i) stf[i] = 0;
  sc.st = stf;
#endif
  unsigned long long s0 = __builtin_amdgcn_s_memtime();
  SolveResult sr = solve_cone_impl<SoloCtx<32, 4>, true, false>(sc, v, w, 100, 1e-11);
  unsigned long long s1 = __builtin_amdgcn_s_memtime();
  {  // the general one-wave solver on the same instance: same minimum, comparable iteration count
    double fl = sr.f; int il = sr.iters;
    SolveView vg = v;
    uint8_t* lflag = ar.get<uint8_t>(40); uint32_t* llist = ar.get<uint32_t>(40);
    for (int i = c.tid(); i < p; i += 64) lflag[i] = (uint8_t)((v.mptr[i + 1] - v.mptr[i]) > kLongRow ? 1 : 0);
    c.sync();
    vg.nlong = (int)c.compact_nonzero_u8(lflag, p, llist); vg.longrow = llist;
    c.sync();
    SolveResult sg = solve_cone_impl<WaveCtx, true, false>(c, vg, w, 100, 1e-11);
    if (threadIdx.x == 0 && (fabs(sg.f - fl) > 1e-9 * (1.0 + fabs(sg.f)) || sg.iters != il || sr.status != 0))
      printf("block %d: lite f %.12e iters %d status %d | general f %.12e iters %d\n", b, fl, il, sr.status, sg.f, sg.iters);
  }
  // cross-check: lite vs general results
  lite_gradient(sc, L, p, rc, g); double gl = threadIdx.x < p ? g[threadIdx.x] : 0.0;
  gradient<SoloCtx<32, 4>, true>(sc, v, rc, g); double gg = threadIdx.x < p ? g[threadIdx.x] : 0.0;
  double err = sc.reduce_max(fabs(gl - gg));
  lite_gather(sc, L, d, y, theta, -1.0, res); double r1 = threadIdx.x < d ? res[threadIdx.x] : 0.0;
  gather_mt<SoloCtx<32, 4>, true>(sc, v, y, theta, -1.0, res); double r2 = threadIdx.x < d ? res[threadIdx.x] : 0.0;
  double err2 = sc.reduce_max(fabs(r1 - r2));
  if (threadIdx.x == 0 && (err > 1e-12 || err2 > 1e-12)) printf("MISMATCH block %d: gradient %.3e gather %.3e\n", b, err, err2);
#ifdef CAVE_STAMPS
  if (threadIdx.x == 0) for (int i = 0; i < 32; ++i) out2[b * 32 + i] = stf[i];
#endif
  if (threadIdx.x == 0) {
    out[b * NSLOT + 0] = t[NSLOT] - t[0];
    for (int i = 1; i < 8; ++i) out[b * NSLOT + i] = (t[i + 1] - t[i]) / reps;
    out[b * NSLOT + 8] = p; out[b * NSLOT + 9] = L.chn8; out[b * NSLOT + 10] = L.cmax; out[b * NSLOT + 11] = v.mptr[p];
    out[b * NSLOT + 12] = th0 / reps; out[b * NSLOT + 13] = th1 / reps;
    out[b * NSLOT + 17] = ti0 / reps; out[b * NSLOT + 18] = ti1 / reps;
    out[b * NSLOT + 14] = s1 - s0; out[b * NSLOT + 15] = sr.iters; out[b * NSLOT + 16] = (s1 - s0) / (sr.iters > 0 ? sr.iters : 1);
  }
}

int main(int argc, char** argv) {
  // input: a binary file written by tools/micro/make_tsp20.py: int32 B, m, d then ctrs [B,m,d] f32, pred [B,d] f32
  FILE* f = fopen(argc > 1 ? argv[1] : "tsp20.bin", "rb");
  if (!f) { printf("no input\n"); return 1; }
  int hdr[3]; fread(hdr, 4, 3, f);
  int B = hdr[0], m = hdr[1], d = hdr[2];
  std::vector<float> ctrs((size_t)B * m * d), pred((size_t)B * d);
  fread(ctrs.data(), 4, ctrs.size(), f); fread(pred.data(), 4, pred.size(), f); fclose(f);
  float *dc, *dp; unsigned long long* dout;
  hipMalloc(&dc, ctrs.size() * 4); hipMalloc(&dp, pred.size() * 4); hipMalloc(&dout, (size_t)B * NSLOT * 8); unsigned long long* dout2; hipMalloc(&dout2, (size_t)B * 32 * 8); hipMemset(dout2, 0, (size_t)B * 32 * 8);
  hipMemcpy(dc, ctrs.data(), ctrs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dp, pred.data(), pred.size() * 4, hipMemcpyHostToDevice);
  uint32_t lds = 64 * 1024, cap = 2600;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k, dim3(B), dim3(64), lds, 0, dc, dp, m, d, cap, lds, dout, 20, dout2);
  hipDeviceSynchronize();
  std::vector<unsigned long long> o((size_t)B * NSLOT);
  hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
  const char* names[] = {"lite_build", "lite_gradient", "lite_gather (residual)", "refresh_clipped", "gj_solve (LOWER)",
                         "general gradient (CSR, 4 lanes/row)", "general gather_mt (CSC)", "reduce_sum + reduce_max", "p", "chn", "cmax", "nnz", "lite_hessian, every weight changes", "lite_hessian, no weight changes", "WHOLE SOLVE cycles", "  Newton iterations", "  cycles per iteration", "lite_gradient between other phases", "lite_gather between other phases"};
  for (int s = 0; s < NSLOT; ++s) {
    double sum = 0, mx = 0; for (int b = 0; b < B; ++b) { sum += o[(size_t)b * NSLOT + s]; mx = mx > o[(size_t)b * NSLOT + s] ? mx : o[(size_t)b * NSLOT + s]; }
    printf("%-40s mean %9.1f  max %9.0f\n", names[s], sum / B, mx);
  }
#ifdef CAVE_STAMPS
  std::vector<unsigned long long> o2((size_t)B * 32);
  hipMemcpy(o2.data(), dout2, o2.size() * 8, hipMemcpyDeviceToHost);
  const char* sn[32] = {"", "", "pgn loop + reduce + test", "hessian", "rhs / moved / misc", "solve_spd", "ls: q", "ls: exact_step", "ls: residual update, f", "",
                        "", "", "", "", "", "", "gradient", "zig-zag / told copy", "attempt setup", "ratio test + reduce", "inner update loop", "psi0 / amax + reduces", "loop top (before gradient)"};
  double its = 0; for (int b = 0; b < B; ++b) its += o[(size_t)b * NSLOT + 15];
  for (int sidx = 0; sidx < 32; ++sidx) {
    double sum = 0; for (int b = 0; b < B; ++b) sum += o2[(size_t)b * 32 + sidx];
    if (sum > 0) printf("  stamp %2d %-28s %9.0f cycles per iteration\n", sidx, sn[sidx] ? sn[sidx] : "", sum / its);
  }
#endif
  return 0;
}
