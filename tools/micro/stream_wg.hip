// Micro-benchmark: how fast can ONE 4-wave workgroup per compute unit stream its instance (TSP-20: 178 KB of dense
// cones) out of HBM?  The pack half of the fused step kernel runs at that occupancy beside the solve waves; its scan
// took 22-27 us per workgroup, the same with 6 rounds of 32 KB and with 3 rounds of 64 KB in flight.
//   variants: loads in flight per lane (U per batch, double-buffered), nt vs plain loads, grid 256 / 512 / 1024
//   hipcc --offload-arch=gfx950 -O3 -o stream_wg stream_wg.hip && ./stream_wg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256, 2) void stream(const float* A, uint32_t n4, float* out, unsigned long long* cyc) {
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  const v4f* A4 = reinterpret_cast<const v4f*>(A) + (size_t)blockIdx.x * n4;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t round = 64u * U * 4u, woff = wave * 64u * U + lane;
  v4f bufA[U], bufB[U];
  float acc = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  auto load = [&](v4f* buf, uint32_t r0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t i = r0 + woff + u * 64u;
      const v4f* p = &A4[i < n4 ? i : n4 - 1u];
      buf[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
  };
  auto use = [&](const v4f* buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc += buf[u].x + buf[u].y + buf[u].z + buf[u].w;
  };
  load(bufA, 0);
  for (uint32_t r0 = 0; r0 < n4; r0 += 2u * round) {
    if (r0 + round < n4) load(bufB, r0 + round);
    use(bufA);
    __syncthreads();
    if (r0 + round < n4) {
      if (r0 + 2u * round < n4) load(bufA, r0 + 2u * round);
      use(bufB);
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 123.456f) out[0] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const uint32_t m = 232, d = 190, n4 = m * d / 4;  // 11020 float4 = 176 KB per instance
  const int NI = 4096;
  float* A;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&A, (size_t)NI * n4 * 16);
  hipMemset(A, 0, (size_t)NI * n4 * 16);
  hipMalloc(&out, 64);
  hipMalloc(&cyc, 8 * NI);
  std::vector<unsigned long long> h(NI);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int grid) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      const float* base = A + (size_t)((rep * 1024) % (NI - grid + 1)) * n4 * 4;  // fresh instances per repeat
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 30720, 0, base, n4, out, cyc);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = std::min(best, ms);
    }
    hipMemcpy(h.data(), cyc, 8 * grid, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + grid);
    printf("%-34s grid %4d: kernel %6.1f us (%.2f TB/s); per-workgroup cycles min %llu median %llu max %llu\n", name, grid, best * 1e3,
           (double)grid * n4 * 16 / (best * 1e-3) / 1e12, h[0], h[grid / 2], h[grid - 1]);
  };
  for (int grid : {256, 512, 1024}) {
    run("U=4  x2 batches, nt", stream<4, true>, grid);
    run("U=8  x2 batches, nt", stream<8, true>, grid);
    run("U=16 x2 batches, nt", stream<16, true>, grid);
    run("U=22 x2 batches (all upfront), nt", stream<22, true>, grid);
    run("U=8  x2 batches, plain", stream<8, false>, grid);
    run("U=22 x2 batches, plain", stream<22, false>, grid);
  }
  return 0;
}
