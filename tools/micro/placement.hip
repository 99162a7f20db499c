// Micro-benchmark: where does the dispatcher put the waves of a "solve-then-pack" grid?
//
// The fused step kernel (cave_amd/csrc/kernels.h cone_step_kernel) launches B one-wave solve instances and the
// 4-wave pack workgroups of the next batch in ONE grid of 256-thread workgroups: solve blocks come first in the
// index order, three of their four waves exit at once.  That only pays if (a) the long-running waves of the solve
// blocks end up spread over the SIMDs of the machine (one per SIMD at B = 1024), and (b) the pack blocks start
// while the solve waves are still running.  This program measures both with stand-in waves that spin on dependent
// FMAs for a given time, 256 VGPRs allocated (the real kernels' budget), the real kernels' LDS sizes.
//
//   hipcc --offload-arch=gfx950 -O3 -o placement placement.hip && ./placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>

struct Rec {
  unsigned hw, xcc;
  unsigned long long t0, t1;
};

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz

__device__ __forceinline__ double spin(double a, unsigned long long ticks) {
  const unsigned long long t0 = now();
  while (now() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 32; ++i) a = fma(a, 0.999, 1e-3);
  }
  return a;
}

// MODE 0: solve wave = wave 0;  1: wave (blockIdx & 3);  2: ticket per CU (global counters keyed by HW_ID)
template <int MODE>
__global__ __launch_bounds__(256, 2) void fused(Rec* rec, double* out, int nsolve, unsigned long long solve_ticks,
                                                unsigned long long pack_ticks, unsigned* tickets) {
  extern __shared__ unsigned char smem[];
  asm volatile("v_mov_b32 v255, 0" ::: "v255");  // the real kernels allocate 256 VGPRs
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  const unsigned long long t0 = now();
  double a = 1.0 + lane;
  const int b = blockIdx.x;
  if (b < nsolve) {
    int sel = 0;
    if (MODE == 1) sel = b & 3;
    if (MODE == 2) {
      // one ticket per workgroup from the counter of its CU; the wave sitting on SIMD (ticket % 4) solves
      unsigned* sh = reinterpret_cast<unsigned*>(smem);
      if (threadIdx.x == 0) {
        const unsigned cu = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
        sh[0] = atomicAdd(&tickets[cu], 1u);
        sh[1] = 0xffffffffu;
      }
      __syncthreads();
      const unsigned want = sh[0] & 3u;
      if (lane == 0 && ((hw >> 4) & 3u) == want) atomicMin(&sh[1], (unsigned)wave);
      __syncthreads();
      sel = sh[1] == 0xffffffffu ? 0 : (int)sh[1];
    }
    if (wave != sel) return;
    a = spin(a, solve_ticks);
  } else {
    a = spin(a, pack_ticks);
  }
  if (lane == 0) {
    Rec r{hw, xcc, t0, now()};
    rec[b * 4 + ((b < nsolve) ? 0 : wave)] = r;
  }
  if (a == 123.456) out[0] = a;
}


// 128-thread blocks: a solve block has two waves (on two SIMDs, normally); it CLAIMS a SIMD of its compute unit with an
// atomicOr on a per-CU bit mask -- the SIMD of wave 0 if free, else that of wave 1 -- and the wave sitting there stays.
// The bit is released when the solve wave ends.  Pack blocks: two waves each.
__global__ __launch_bounds__(128, 2) void fused2(Rec* rec, double* out, int nsolve, unsigned long long solve_ticks,
                                                 unsigned long long pack_ticks, unsigned* masks) {
  extern __shared__ unsigned char smem[];
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  const unsigned long long t0 = now();
  double a = 1.0 + lane;
  const int b = blockIdx.x;
  const unsigned cu = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
  if (b < nsolve) {
    unsigned* sh = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) sh[2 + wave] = (hw >> 4) & 3u;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned s0 = sh[2], s1 = sh[3];
      unsigned sel = 0;
      const unsigned old0 = atomicOr(&masks[cu], 1u << s0);
      if (old0 & (1u << s0)) {
        const unsigned old1 = (s1 != s0) ? atomicOr(&masks[cu], 1u << s1) : 0xfu;
        if (!(old1 & (1u << s1))) sel = 1; else sel = 2;  // 2: conflict, wave 0 stays without a claim
      }
      sh[0] = sel;
    }
    __syncthreads();
    const unsigned sel = sh[0];
    const int stay = sel == 1 ? 1 : 0;
    if (wave != stay) return;
    a = spin(a, solve_ticks);
    if (lane == 0 && sel != 2) atomicAnd(&masks[cu], ~(1u << ((hw >> 4) & 3u)));
    if (lane == 0) rec[b * 4] = Rec{hw | (sel == 2 ? 0x80000000u : 0u), xcc, t0, now()};
  } else {
    a = spin(a, pack_ticks);
    if (lane == 0) rec[b * 4 + wave] = Rec{hw, xcc, t0, now()};
  }
  if (a == 123.456) out[0] = a;
}

__global__ __launch_bounds__(64, 2) void solo(Rec* rec, double* out, unsigned long long ticks) {
  extern __shared__ unsigned char smem[];
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  const unsigned long long t0 = now();
  double a = spin(1.0 + threadIdx.x, ticks);
  if (threadIdx.x == 0) rec[blockIdx.x * 4] = Rec{hw, xcc, t0, now()};
  if (a == 123.456) out[0] = a;
}

static void report(const char* name, const std::vector<Rec>& r, int nsolve, int npack, int pw = 4) {
  std::map<unsigned, int> per_simd, per_cu;
  unsigned long long tmin = ~0ull, solve_end = 0, pack_first = ~0ull, pack_end = 0;
  for (int b = 0; b < nsolve; ++b) tmin = std::min(tmin, r[b * 4].t0);
  for (int b = 0; b < nsolve; ++b) {
    const Rec& x = r[b * 4];
    const unsigned cu = ((x.xcc & 15u) << 8) | (((x.hw >> 13) & 7u) << 5) | (((x.hw >> 12) & 1u) << 4) | ((x.hw >> 8) & 15u);
    per_cu[cu]++;
    per_simd[(cu << 2) | ((x.hw >> 4) & 3u)]++;
    solve_end = std::max(solve_end, x.t1);
  }
  for (int b = nsolve; b < nsolve + npack; ++b)
    for (int w = 0; w < pw; ++w) {
      pack_first = std::min(pack_first, r[b * 4 + w].t0);
      pack_end = std::max(pack_end, r[b * 4 + w].t1);
    }
  int hist_cu[16] = {0}, hist_simd[16] = {0};
  for (auto& kv : per_cu) hist_cu[std::min(kv.second, 15)]++;
  for (auto& kv : per_simd) hist_simd[std::min(kv.second, 15)]++;
  printf("%-28s CUs used %3zu  solve waves per CU histogram [1..6]: %d %d %d %d %d %d | per SIMD [1..4]: %d %d %d %d (SIMDs used %zu)\n",
         name, per_cu.size(), hist_cu[1], hist_cu[2], hist_cu[3], hist_cu[4], hist_cu[5], hist_cu[6], hist_simd[1],
         hist_simd[2], hist_simd[3], hist_simd[4], per_simd.size());
  printf("%-28s solve end %.1f us", "", (solve_end - tmin) / 100.0);
  if (npack) printf("; first pack wave starts %.1f us, last pack wave ends %.1f us", (pack_first - tmin) / 100.0, (pack_end - tmin) / 100.0);
  printf("\n");
}

int main(int argc, char** argv) {
  const int nsolve = 1024, npack = 1024;
  const unsigned lds = argc > 1 ? (unsigned)atoi(argv[1]) : 31488u;
  Rec* drec;
  double* dout;
  unsigned* dt;
  hipMalloc(&drec, sizeof(Rec) * 4 * (nsolve + npack));
  hipMalloc(&dout, 64);
  hipMalloc(&dt, 4 * 4096);
  std::vector<Rec> h(4 * (nsolve + npack));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const unsigned long long solve_ticks = 11000, pack_ticks = 2800;  // 110 us, 28 us
  auto run = [&](const char* name, auto launch, int np) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(drec, 0, sizeof(Rec) * h.size());
      hipMemset(dt, 0, 4 * 4096);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipDeviceSynchronize();
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), drec, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
    printf("%-28s kernel %.1f us\n", name, ms * 1e3);
    report(name, h, nsolve, np);
  };
  hipFuncSetAttribute((const void*)solo, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)fused<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)fused<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)fused<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("LDS per workgroup: %u bytes; solve stand-in %.0f us, pack stand-in %.0f us per workgroup\n", lds, solve_ticks / 100.0,
         pack_ticks / 100.0);
  run("solo 64-thread blocks (33000 B)", [&] { hipLaunchKernelGGL(solo, dim3(nsolve), dim3(64), 33000, 0, drec, dout, solve_ticks); }, 0);
  run("fused, solve only, wave 0", [&] { hipLaunchKernelGGL(fused<0>, dim3(nsolve), dim3(256), lds, 0, drec, dout, nsolve, solve_ticks, pack_ticks, dt); }, 0);
  run("fused, solve only, wave b&3", [&] { hipLaunchKernelGGL(fused<1>, dim3(nsolve), dim3(256), lds, 0, drec, dout, nsolve, solve_ticks, pack_ticks, dt); }, 0);
  run("fused, solve only, ticket", [&] { hipLaunchKernelGGL(fused<2>, dim3(nsolve), dim3(256), lds, 0, drec, dout, nsolve, solve_ticks, pack_ticks, dt); }, 0);
  run("fused solve+pack, wave 0", [&] { hipLaunchKernelGGL(fused<0>, dim3(nsolve + npack), dim3(256), lds, 0, drec, dout, nsolve, solve_ticks, pack_ticks, dt); }, npack);
  run("fused solve+pack, wave b&3", [&] { hipLaunchKernelGGL(fused<1>, dim3(nsolve + npack), dim3(256), lds, 0, drec, dout, nsolve, solve_ticks, pack_ticks, dt); }, npack);
  run("fused solve+pack, ticket", [&] { hipLaunchKernelGGL(fused<2>, dim3(nsolve + npack), dim3(256), lds, 0, drec, dout, nsolve, solve_ticks, pack_ticks, dt); }, npack);
  {  // 128-thread blocks, claim election, two-wave pack stand-ins of 57 us, 26.5 KB of LDS (six workgroups per CU)
    hipFuncSetAttribute((const void*)fused2, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(drec, 0, sizeof(Rec) * h.size());
      hipMemset(dt, 0, 4 * 4096);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(fused2, dim3(nsolve + npack), dim3(128), 26624, 0, drec, dout, nsolve, solve_ticks, 5700ull, dt);
      hipEventRecord(e1);
      hipDeviceSynchronize();
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), drec, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
    int conflicts = 0;
    for (int b = 0; b < nsolve; ++b) conflicts += (h[b * 4].hw >> 31) & 1u;
    for (int b = 0; b < nsolve; ++b) h[b * 4].hw &= 0x7fffffffu;
    printf("%-28s kernel %.1f us; claim conflicts %d\n", "fused2 (128-thread blocks)", ms * 1e3, conflicts);
    report("fused2 (128-thread blocks)", h, nsolve, npack, 2);
  }
  return 0;
}
