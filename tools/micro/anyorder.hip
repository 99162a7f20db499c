// Micro-benchmark: do two kernels launched back to back on ONE stream overlap when the second is launched with
// hipExtAnyOrderLaunch (no barrier bit in its AQL packet)?  (hip_ext.h says the flag is not supported on gfx9.)
//   stand-ins: "solve" = 1024 one-wave workgroups spinning 110 us (256 VGPRs, 25 KB LDS);
//              "pack"  = 1024 four-wave workgroups spinning 40 us (128 VGPRs, 30 KB LDS).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__device__ __forceinline__ double spin(double a, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 32; ++i) a = fma(a, 0.999, 1e-3);
  }
  return a;
}
__global__ __launch_bounds__(64, 2) void solve(double* out, unsigned long long ticks) {
  extern __shared__ unsigned char smem[];
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  double a = spin(1.0 + threadIdx.x, ticks);
  if (a == 123.456) out[0] = a;
}
__global__ __launch_bounds__(256, 4) void pack(double* out, unsigned long long ticks) {
  extern __shared__ unsigned char smem[];
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  double a = spin(1.0 + threadIdx.x, ticks);
  if (a == 123.456) out[0] = a;
}
int main() {
  double* out;
  hipMalloc(&out, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timed = [&](const char* name, auto fn) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      fn();
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-60s %7.1f us\n", name, best * 1e3);
  };
  timed("solve alone", [&] { hipLaunchKernelGGL(solve, dim3(1024), dim3(64), 25344, s, out, 11000ull); });
  timed("pack alone", [&] { hipLaunchKernelGGL(pack, dim3(1024), dim3(256), 30720, s, out, 4000ull); });
  timed("solve; pack (plain launches)", [&] {
    hipLaunchKernelGGL(solve, dim3(1024), dim3(64), 25344, s, out, 11000ull);
    hipLaunchKernelGGL(pack, dim3(1024), dim3(256), 30720, s, out, 4000ull);
  });
  timed("solve; pack with hipExtAnyOrderLaunch", [&] {
    hipLaunchKernelGGL(solve, dim3(1024), dim3(64), 25344, s, out, 11000ull);
    hipExtLaunchKernelGGL(pack, dim3(1024), dim3(256), 30720, s, nullptr, nullptr, hipExtAnyOrderLaunch, out, 4000ull);
  });
  timed("both with hipExtAnyOrderLaunch", [&] {
    hipExtLaunchKernelGGL(solve, dim3(1024), dim3(64), 25344, s, nullptr, nullptr, hipExtAnyOrderLaunch, out, 11000ull);
    hipExtLaunchKernelGGL(pack, dim3(1024), dim3(256), 30720, s, nullptr, nullptr, hipExtAnyOrderLaunch, out, 4000ull);
  });
  return 0;
}
