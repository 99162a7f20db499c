// Micro-benchmark: what does rocprofv3's FETCH_SIZE report for GATHER-type reads on gfx950?
//
// MI355X_MICROARCH.md: FETCH_SIZE counts half of the bytes of a wide coalesced streaming read (16 B / lane; 128-B
// requests tallied at 64 B), so profiles/*_pmc_summary.json double it.  The large-cone kernels gather 2- to 8-byte
// entries (16-bit indices, fp64 operands): VERDICT r3 asked whether the doubling holds there.  Each kernel below reads
// a KNOWN number of distinct bytes out of a buffer far larger than the 256 MB Infinity Cache, once:
//   stream16    16 B / lane, consecutive                       (the guide's case)
//   stream4      4 B / lane, consecutive (256 B per wave instruction)
//   stream2      2 B / lane, consecutive (128 B per wave instruction)
//   gather8_64   8 B / lane, one entry per 64-B line  (every touched line is fetched whole: 64 B per entry)
//   gather8_128  8 B / lane, one entry per 128-B line
//   gather2_32   2 B / lane, one entry per 32 B  (two entries per 64-B line)
// Run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_gather ; tools/micro/fetch_gather_report.py out
// Expected if the counter tallies 64 B per 128-B request only for wide streams: ratio 0.5 for stream16, ~1.0 x the
// bytes of the touched 64-B lines for the gathers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <class T>
__global__ __launch_bounds__(256) void stream(const T* a, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = a[i];
    acc += (float)(reinterpret_cast<const unsigned char*>(&v)[0]);
  }
  if (acc == 12345.f) out[0] = acc;
}

// one T per `stride` bytes
template <class T>
__global__ __launch_bounds__(256) void gather(const unsigned char* a, size_t nent, size_t stride, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nent; i += (size_t)gridDim.x * 256) {
    const T v = *reinterpret_cast<const T*>(a + i * stride);
    acc += (float)(reinterpret_cast<const unsigned char*>(&v)[0]);
  }
  if (acc == 12345.f) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB per pass: four times the Infinity Cache
  unsigned char* buf;
  float* out;
  hipMalloc(&buf, bytes * 6);
  hipMalloc(&out, 64);
  hipMemset(buf, 1, bytes * 6);
  hipDeviceSynchronize();
  const int grid = 256 * 16;
  // each kernel reads its own 1 GiB region (nothing it touches is left in a cache by a predecessor)
  hipLaunchKernelGGL(stream<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)(buf + 0 * bytes), bytes / 16, out);
  hipLaunchKernelGGL(stream<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)(buf + 1 * bytes), bytes / 4, out);
  hipLaunchKernelGGL(stream<uint16_t>, dim3(grid), dim3(256), 0, 0, (const uint16_t*)(buf + 2 * bytes), bytes / 2, out);
  hipLaunchKernelGGL(gather<uint64_t>, dim3(grid), dim3(256), 0, 0, buf + 3 * bytes, bytes / 64, (size_t)64, out);
  hipLaunchKernelGGL(gather<uint64_t>, dim3(grid), dim3(256), 0, 0, buf + 4 * bytes, bytes / 128, (size_t)128, out);
  hipLaunchKernelGGL(gather<uint16_t>, dim3(grid), dim3(256), 0, 0, buf + 5 * bytes, bytes / 32, (size_t)32, out);
  hipDeviceSynchronize();
  printf("fetch_gather: six kernels over 1 GiB regions done\n");
  return 0;
}
