// Micro-benchmark: single-wave cost (cycles per instruction) of the primitives the per-instance solver is built from.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../cave_amd/csrc/wave_prims.h"
using namespace cave;

#define REP8(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP64(...) REP8(REP8(__VA_ARGS__))

template <int WHICH>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, int n) {
  __shared__ double lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = 1.0 + i * 1e-3;
  __syncthreads();
  double a = 1.0 + lane * 1e-3, b = 0.999, c = 1e-3, d = 0.5, e = 0.25, f = 0.125, g = 2.0, h = 3.0;
  int idx = lane;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
    if (WHICH == 0) { REP64(a = fma(a, b, c);) }                                   // dependent f64 fma
    if (WHICH == 1) { REP8(a = fma(a, b, c); d = fma(d, b, c); e = fma(e, b, c); f = fma(f, b, c); g = fma(g, b, c); h = fma(h, b, c); a = fma(a, b, d); e = fma(e, b, f);) }  // mostly independent f64 fma
    if (WHICH == 2) { REP64(a = __builtin_amdgcn_rcp(a);) }                         // dependent v_rcp_f64
    if (WHICH == 3) { REP64(a = fma(a, readlane_f64(a, 3), c);) }                   // readlane x2 + fma, dependent
    if (WHICH == 4) { REP64(a = half_bcast_f64<5>(a);) }                            // dependent swizzle pair
    if (WHICH == 5) { REP8(a = half_bcast_f64<5>(a); d = half_bcast_f64<6>(d); e = half_bcast_f64<7>(e); f = half_bcast_f64<8>(f); g = half_bcast_f64<9>(g); h = half_bcast_f64<10>(h); b = half_bcast_f64<11>(b); c = half_bcast_f64<12>(c);) }  // independent swizzles
    if (WHICH == 6) { REP64(idx = (int)lds[idx & 1023];) ; a += idx; }              // dependent LDS read (f64) + cvt
    if (WHICH == 7) { REP64(a = wave_sum_f64(a) * 1e-2;) }                          // wave_sum chain
    if (WHICH == 8) { REP64(a = from_half_f64<1>(a) + c;) }                         // permlane32 swap pair + add
    if (WHICH == 9) { float x = (float)a; REP64(x = fmaf(x, 0.999f, 1e-3f);) a = x; } // dependent f32 fma
    if (WHICH == 10) { REP64(a = dpp_f64<0x111, 0xf>(0.0, a) + c;) }                // dpp mov pair + add
    if (WHICH == 11) { REP64(__hip_atomic_fetch_add(&lds[(lane * 7 + 3) & 1023], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);) }  // LDS f64 atomic, conflict-free
    if (WHICH == 12) { REP64(__hip_atomic_fetch_add(&lds[lane & 3], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);) }  // 16-way conflicting
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = a + b + c + d + e + f + g + h + idx + lds[lane];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int W> void run(const char* name, double* dout, unsigned long long* dc, int nb) {
  int n = 20;
  hipLaunchKernelGGL(k<W>, dim3(nb), dim3(64), 0, 0, dout, dc, n);
  hipLaunchKernelGGL(k<W>, dim3(nb), dim3(64), 0, 0, dout, dc, n);
  hipDeviceSynchronize();
  std::vector<unsigned long long> c(nb);
  hipMemcpy(c.data(), dc, nb * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : c) s += x;
  printf("%-52s %7.1f cycles per step\n", name, s / nb / n / 64);
}
int main(int argc, char** argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 1024;  // 1024 blocks of one wave = one wave per SIMD
  double* dout; unsigned long long* dc;
  hipMalloc(&dout, nb * 64 * 8); hipMalloc(&dc, nb * 8);
  run<0>("dependent v_fma_f64", dout, dc, nb);
  run<1>("independent v_fma_f64 (8 chains)", dout, dc, nb);
  run<2>("dependent v_rcp_f64", dout, dc, nb);
  run<3>("2 v_readlane + v_fma_f64 (dependent)", dout, dc, nb);
  run<4>("dependent ds_swizzle pair (f64 bcast)", dout, dc, nb);
  run<5>("independent ds_swizzle pair", dout, dc, nb);
  run<6>("dependent LDS read f64 + cvt", dout, dc, nb);
  run<7>("wave_sum_f64 chain", dout, dc, nb);
  run<8>("permlane32_swap pair + add (dependent)", dout, dc, nb);
  run<9>("dependent v_fma_f32", dout, dc, nb);
  run<10>("dpp mov pair + add_f64 (dependent)", dout, dc, nb);
  run<11>("LDS atomic add f64, conflict-free (issue)", dout, dc, nb);
  run<12>("LDS atomic add f64, 16 lanes per address (issue)", dout, dc, nb);
  return 0;
}
