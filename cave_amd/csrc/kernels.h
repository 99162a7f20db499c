// kernels.h — the gfx950 kernel templates and the host-side launch table.
//
// One workgroup of 1, 2 or 4 cooperating 64-lane wavefronts per training instance (grid = B; the
// `waves` argument of the C ABI picks the shape, see include/cave_hip.h):
//   1. stream the instance's dense (m_max x d) block from HBM with 16-byte loads, two batches in
//      flight per wave, keeping only the non-zeros (ordered CSR in LDS);
//   2. classify rows / pair equalities / build CSC            (cone_core.h build_cone)
//   3. projected semismooth Newton in fp64, Newton systems solved in registers (solve_cone)
//   4. fused epilogue: proj, rnorm, loss target, loss, d loss / d pred.
// Cones beyond LDS run on persistent workgroups over a global workspace (the *_large kernels).
// Instances are independent, so the block->instance map is the identity and no
// XCD-aware remap is needed (nothing is shared through L2).
//
// Every kernel shape is instantiated in its own translation unit (k_*.hip: one explicit
// instantiation + its launch function), so the library builds in parallel and a change to one
// kernel recompiles one file; cave_hip.hip holds the C ABI and calls the launch functions below.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cave_hip.h"
#include "cone_common.h"
#include "cone_core.h"
#include "ctx_wave.h"
#include "ctx_block.h"
#include "cone_instance.h"

namespace cave {

#ifdef CAVE_STAMPS
// diagnostic (unity) build only: one translation unit, one buffer
static __device__ unsigned long long g_stamp_buf[16 * 8192];
#endif

using Ctx1 = WaveCtx;      // one wave per instance, reduced systems up to 64 rows
using Ctx4 = BlockCtx<4>;  // 4-wave workgroup per instance, reduced systems up to 32 rows
using Ctx2 = BlockCtx<2>;  // 2-wave workgroup per instance
using CtxW = BlockCtx<4, true>;  // 4 waves with the full register budget: for launches whose LDS arena allows one
                                 // workgroup per CU anyway (TSP-50: 100-160 KB); reduced systems up to 64 rows

// launch bounds: NT threads; for the 4-wave context ask for 4 waves per SIMD (= 4 workgroups per CU,
// the residency LDS allows), which caps the kernel at 128 VGPRs
#define CAVE_BOUNDS(C) __launch_bounds__(C::NT, C::MIN_WAVES_PER_EU)

template <class C>
__global__ CAVE_BOUNDS(C) void cone_dense_kernel(DenseParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  // one workgroup per instance (grid = B): no loop, so nothing loop-invariant is hoisted into long live ranges
  const int64_t b = blockIdx.x;
  if (b >= P.B) return;
#ifdef CAVE_STAMPS
  for (int i = 0; i < 32; ++i) c.st[i] = 0;
  unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  run_dense_instance(c, smem, P, b);
#ifdef CAVE_STAMPS
  c.st[14] = __builtin_amdgcn_s_memtime() - mt0;
  c.st[15] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz
  if (c.tid() == 0 && b < 8192) for (int i = 0; i < 16; ++i) g_stamp_buf[b * 16 + i] = c.st[i];
#endif
}

template <class C>
__global__ CAVE_BOUNDS(C) void cone_pack_kernel(PackParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  const int64_t b = blockIdx.x;
#ifdef CAVE_STAMPS
  for (int i = 0; i < 32; ++i) c.st[i] = 0;
  unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (b < P.B) run_pack_instance(c, smem, P, b);
#ifdef CAVE_STAMPS
  c.st[14] = __builtin_amdgcn_s_memtime() - mt0;
  c.st[15] = rt0;                                // absolute start / end (100 MHz): launch skew across workgroups
  c.st[9] = __builtin_amdgcn_s_memrealtime();
  if (c.tid() == 0 && b < 8192 && b < P.B) for (int i = 0; i < 16; ++i) g_stamp_buf[b * 16 + i] = c.st[i];
#endif
}

template <class C>
__global__ CAVE_BOUNDS(C) void cone_packed_kernel(PackedParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  const int64_t b = blockIdx.x;
#ifdef CAVE_STAMPS
  for (int i = 0; i < 32; ++i) c.st[i] = 0;
  unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (b < P.B) run_packed_instance(c, smem, P, b);
#ifdef CAVE_STAMPS
  c.st[14] = __builtin_amdgcn_s_memtime() - mt0;
  c.st[15] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz
  c.st[0] = rt0;                                      // absolute start
  if (c.tid() == 0 && b < 8192 && b < P.B) for (int i = 0; i < 16; ++i) g_stamp_buf[b * 16 + i] = c.st[i];
#endif
}

// ---- large-cone path (cone_band.h): persistent workgroups, arena = a slice of a global
// workspace, LDS = "hot" arena for the small per-iteration arrays.  256 VGPRs (2 waves / SIMD).
struct LargeWs {
  unsigned char* base;
  uint64_t slice;  // bytes per workgroup (< 4 GiB)
};
using CtxL = BlockCtx<4, true>;
using CtxL2 = BlockCtx<2, true>;

template <class C>
__global__ __launch_bounds__(C::NT, 2) void cone_dense_large_kernel(DenseParams P, LargeWs W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  unsigned char* ws = W.base + (uint64_t)blockIdx.x * W.slice;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
    run_dense_instance<C, true>(c, smem, P, b, ws, (uint32_t)W.slice);
    __syncthreads();
  }
}

template <class C>
__global__ __launch_bounds__(C::NT, 2) void cone_pack_large_kernel(PackParams P, LargeWs W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  unsigned char* ws = W.base + (uint64_t)blockIdx.x * W.slice;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
    run_pack_instance<C, true>(c, smem, P, b, ws, (uint32_t)W.slice);
    __syncthreads();
  }
}

// (MINB = waves per SIMD the register budget is set for: 2 -> 256 VGPRs.  C = CtxL: 4 waves, two workgroups per CU;
//  CtxL2 / Ctx1: 2 / 1 waves, four and more workgroups per CU where the LDS allows -- for batches that fill the
//  chip several times over with narrow-band cones, whose elimination runs on one wave anyway: cone_band.h)
template <class C, int MINB>
__global__ __launch_bounds__(C::NT, MINB) void cone_packed_large_kernel(PackedParams P, LargeWs W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  unsigned char* ws = W.base + (uint64_t)blockIdx.x * W.slice;
  for (int64_t b = blockIdx.x; b < P.B; b += gridDim.x) {
#ifdef CAVE_STAMPS
    for (int i = 0; i < 32; ++i) c.st[i] = 0;
    unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    run_packed_large_instance<C>(c, smem, P, b, ws, (uint32_t)W.slice);
#ifdef CAVE_STAMPS
    c.st[14] = __builtin_amdgcn_s_memtime() - mt0;
    c.st[15] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz
    if (c.tid() == 0 && b < 8192) for (int i = 0; i < 16; ++i) g_stamp_buf[b * 16 + i] = c.st[i];
#endif
    __syncthreads();
  }
}

// ---------------------------------------------------------------- launch table (host)
// One function per kernel shape, defined next to its instantiation (k_*.hip).  Each sets the dynamic-LDS
// attribute when the launch needs more than 48 KiB, launches on `stream` and returns hipGetLastError().
#define CAVE_DECL_LAUNCH(NAME, PARAMS) hipError_t NAME(unsigned grid, uint32_t lds, hipStream_t stream, const PARAMS& P)
#define CAVE_DECL_LAUNCH_LARGE(NAME, PARAMS) \
  hipError_t NAME(unsigned grid, uint32_t lds, hipStream_t stream, const PARAMS& P, const LargeWs& W)

CAVE_DECL_LAUNCH(launch_dense_w1, DenseParams);
CAVE_DECL_LAUNCH(launch_dense_w2, DenseParams);
CAVE_DECL_LAUNCH(launch_dense_w4, DenseParams);
CAVE_DECL_LAUNCH(launch_dense_w8, DenseParams);
CAVE_DECL_LAUNCH(launch_pack_w1, PackParams);
CAVE_DECL_LAUNCH(launch_pack_w2, PackParams);
CAVE_DECL_LAUNCH(launch_pack_w4, PackParams);
CAVE_DECL_LAUNCH(launch_pack_w8, PackParams);
CAVE_DECL_LAUNCH(launch_packed_w1, PackedParams);
CAVE_DECL_LAUNCH(launch_packed_w2, PackedParams);
CAVE_DECL_LAUNCH(launch_packed_w4, PackedParams);
CAVE_DECL_LAUNCH(launch_packed_w8, PackedParams);
CAVE_DECL_LAUNCH_LARGE(launch_dense_large, DenseParams);
CAVE_DECL_LAUNCH_LARGE(launch_pack_large, PackParams);
CAVE_DECL_LAUNCH_LARGE(launch_packed_large_w1, PackedParams);
CAVE_DECL_LAUNCH_LARGE(launch_packed_large_w2, PackedParams);
CAVE_DECL_LAUNCH_LARGE(launch_packed_large_w4, PackedParams);

template <class K>
static inline hipError_t ensure_lds(K kernel, uint32_t bytes) {
  if (bytes <= 48u * 1024u) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

// body of a launch function: KERNEL is a fully specialised kernel, NT its workgroup size
#define CAVE_DEFINE_LAUNCH(NAME, PARAMS, KERNEL, NT)                                                \
  CAVE_DECL_LAUNCH(NAME, PARAMS) {                                                                  \
    hipError_t e = ensure_lds(KERNEL, lds);                                                         \
    if (e != hipSuccess) return e;                                                                  \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(NT), (size_t)lds, stream, P);                       \
    return hipGetLastError();                                                                       \
  }
#define CAVE_DEFINE_LAUNCH_LARGE(NAME, PARAMS, KERNEL, NT)                                          \
  CAVE_DECL_LAUNCH_LARGE(NAME, PARAMS) {                                                            \
    hipError_t e = ensure_lds(KERNEL, lds);                                                         \
    if (e != hipSuccess) return e;                                                                  \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(NT), (size_t)lds, stream, P, W);                    \
    return hipGetLastError();                                                                       \
  }

}  // namespace cave
