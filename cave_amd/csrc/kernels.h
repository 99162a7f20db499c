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
#include "cone_step.h"

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

// ---- fused step kernel (cone_step.h): blocks [0, S.B) solve the current batch from its lite store, one wave each;
// blocks [S.B, S.B + Q.B) pack the next dense batch, CP::NWAVES waves each.  ONE workgroup shape (CP::NT threads, 256
// VGPRs: the solver's budget) and one LDS size for both halves.
//
// Why TWO-wave workgroups.  A compute unit holds eight 256-register waves, two per SIMD.  Four of them are the solve
// waves of its four solve blocks; the pack workgroups get the other four -- and a workgroup is only dispatched when
// ALL its waves find a slot.  With four-wave pack workgroups a second one could not start before the LAST solve wave
// of the compute unit had ended: the pack ran strictly one workgroup (~40 us) at a time per compute unit and trailed
// the solve by 35 us (150 us per step).  Two-wave workgroups go two at a time from the start, a third as soon as two
// solve waves are done (tools/micro/placement.hip, stand-in waves: 120 us for 110 us of solve).
//
// Wave election of a solve block.  All but one of its waves exit at once; the one that stays must not share its SIMD
// with the staying wave of another solve block of the same compute unit (two 256-register waves fill a SIMD: pack
// waves could not be placed there, and the two solves would halve each other's issue rate -- "wave 0 stays" leaves 6 %
// of the SIMDs with two solve waves and the stand-in launch takes 171 us instead of 120).  So a solve block CLAIMS a
// SIMD of its compute unit (HW_REG_HW_ID / XCC_ID) in a per-compute-unit bit mask in global memory: the SIMD of its
// wave 0 if free, else that of wave 1, ...; the wave sitting on the claimed SIMD stays and releases the bit when it
// ends.  (Measured: 1024 solve waves on 1024 different SIMDs, no failed claim.)
template <int NW>
__device__ __forceinline__ int step_elect_wave(unsigned char* smem, uint32_t* masks, uint32_t& claim) {
  uint32_t* sh = reinterpret_cast<uint32_t*>(smem);
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: SIMD [5:4], CU [11:8], SH [12], SE [15:13]
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID [3:0]
  const uint32_t cu = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
  if ((threadIdx.x & 63u) == 0u) sh[4 + (threadIdx.x >> 6)] = (hw >> 4) & 3u;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t sel = 0, bit = 0;
    for (int w = 0; w < NW; ++w) {
      const uint32_t sb = 1u << sh[4 + w];
      const uint32_t old = atomicOr(&masks[cu], sb);
      if (!(old & sb)) { sel = (uint32_t)w; bit = sb; break; }
    }
    sh[0] = sel;   // (no claim succeeded: wave 0 stays without one)
    sh[1] = bit;
    sh[2] = cu;
  }
  __syncthreads();
  claim = sh[1] ? ((sh[2] << 4) | sh[1]) : 0u;
  return (int)sh[0];
}
__device__ __forceinline__ void step_release(uint32_t* masks, uint32_t claim) {
  if (claim) atomicAnd(&masks[claim >> 4], ~(claim & 15u));
}

#ifndef CAVE_STEP_PACK_PRIO
#define CAVE_STEP_PACK_PRIO 1   // wave priorities of the two halves (A/B builds: tools/diag/build_variant.sh)
#endif
#ifndef CAVE_STEP_SOLVE_PRIO
#define CAVE_STEP_SOLVE_PRIO 0
#endif
template <class CP>
__global__ __launch_bounds__(CP::NT, 2) void cone_step_kernel(StepParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int64_t b = blockIdx.x;
  if (b < P.S.B) {
    uint32_t claim = 0;
    const int sel = step_elect_wave<CP::NWAVES>(smem, P.tickets, claim);
    if ((int)(threadIdx.x >> 6) != sel) return;
    if (CAVE_STEP_SOLVE_PRIO) __builtin_amdgcn_s_setprio(CAVE_STEP_SOLVE_PRIO);
    SoloCtx<32, 4> sc;
    sc.lane = (int)(threadIdx.x & 63u);
#ifdef CAVE_STAMPS
    unsigned long long stamps[32];
    for (int i = 0; i < 32; ++i) stamps[i] = 0;
    sc.st = stamps;
    unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    run_lite_instance(sc, smem + kStepElectBytes, P.lds_bytes - kStepElectBytes, P.S, b);
    if (sc.lane == 0) step_release(P.tickets, claim);
#ifdef CAVE_STAMPS
    stamps[14] = __builtin_amdgcn_s_memtime() - mt0;
    stamps[15] = __builtin_amdgcn_s_memrealtime() - rt0;  // 100 MHz
    stamps[0] = rt0;
    if (sc.lane == 0 && b < 4096) for (int i = 0; i < 16; ++i) g_stamp_buf[b * 16 + i] = stamps[i];
#endif
    return;
  }
  // the pack half's second round of workgroups starts when the first ends: its waves issue ahead of the solve wave they
  // share a SIMD with (k_step.hip has the measurements)
#ifdef CAVE_STEP_PACK_PRIO_FIRST   // (A/B) only the first CAVE_STEP_PACK_PRIO_FIRST pack workgroups run at the raised priority
  if ((int64_t)blockIdx.x - P.S.B < CAVE_STEP_PACK_PRIO_FIRST) __builtin_amdgcn_s_setprio(CAVE_STEP_PACK_PRIO);
#else
  if (CAVE_STEP_PACK_PRIO) __builtin_amdgcn_s_setprio(CAVE_STEP_PACK_PRIO);
#endif
  CP c;
  c.init(smem);
  const int64_t q = b - P.S.B;
#ifdef CAVE_STAMPS
  for (int i = 0; i < 32; ++i) c.st[i] = 0;
  unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (q < P.Q.B) run_pack_lite_instance(c, smem, P.lds_bytes, P.Q, q);
#ifdef CAVE_STAMPS
  c.st[14] = __builtin_amdgcn_s_memtime() - mt0;
  c.st[15] = rt0;
  c.st[9] = __builtin_amdgcn_s_memrealtime();
  if (c.tid() == 0 && q < 4096 && q < P.Q.B) for (int i = 0; i < 16; ++i) g_stamp_buf[(4096 + q) * 16 + i] = c.st[i];
#endif
}

// packed cone store -> lite store, one workgroup per instance (run once per store: cones are static)
template <class C>
__global__ CAVE_BOUNDS(C) void lite_from_packed_kernel(LiteFromPackedParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  C c;
  c.init(smem);
  const int64_t b = blockIdx.x;
  if (b < P.n) run_lite_from_packed(c, smem, P, b);
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
CAVE_DECL_LAUNCH(launch_step, StepParams);
CAVE_DECL_LAUNCH(launch_lite_from_packed, LiteFromPackedParams);

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
