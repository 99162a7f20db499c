// ctx_wave.h — SPMD context: ONE 64-lane wavefront per instance (workgroup = one wave).
//
// __syncthreads() lowers to an LDS-counter wait (the s_barrier is elided by hipcc under
// __launch_bounds__(64)); LDS operations of one wave execute in order, which is what makes the
// "rows in sequence, lanes over entries" loops of cone_core.h race-free.  Holds reduced systems of
// up to 64 rows in registers (the 4-wave context of ctx_block.h stops at 32).
#pragma once
#include "wave_prims.h"

namespace cave {

struct WaveCtx {
  static constexpr int NT = 64;
  static constexpr int TEAM = 4;         // lanes sharing one CSR row in gradient()
  static constexpr int SCAN_UNROLL = 8;  // 8 x 1 KiB dwordx4 loads per batch, two batches in flight
  static constexpr int PMAX = 64;        // largest reduced system solve_spd holds in registers
  static constexpr int KREG = 4;         // line search keeps r, q in registers when d <= KREG * NT
  static constexpr uint32_t SCRATCH_BYTES = 0;
  static constexpr int NWAVES = 1;       // waves per instance
  static constexpr bool LITE_OK = true;  // 256-register budget: carries the one-wave lite solver (cone_core.h)
  static constexpr int MIN_WAVES_PER_EU = 2;  // 256 registers: two one-wave workgroups per SIMD (B > 1024 keeps latency hiding)
  static constexpr int WL = 64;          // lanes per wave
  int lane;
#ifdef CAVE_STAMPS
  unsigned long long st[32];  // [0,16) exported per instance; [16,32) scratch slots of the fine stamps
#endif
  __device__ __forceinline__ void init(unsigned char*) { lane = (int)threadIdx.x; }
  __device__ __forceinline__ void broadcast_from_wave0(double&, int&, int&) const {}
  __device__ __forceinline__ int tid() const { return lane; }
  __device__ __forceinline__ int wave_id() const { return 0; }
  __device__ __forceinline__ int lane_id() const { return lane; }
  // sum / max over the lanes of the calling wave only (no barrier; every lane gets the result)
  __device__ __forceinline__ double wave_sum(double v) const { return wave_sum_f64(v); }
  __device__ __forceinline__ double wave_max(double v) const { return wave_max_f64(v); }
  // lane l gets lane l-1's value (lane 0: 0)
  __device__ __forceinline__ double wave_shift_up(double v) const { return dpp_f64<0x138, 0xf>(0.0, v); }  // wave_shr:1
  // barrier that orders LDS traffic only (typed ds_* accesses): does not wait for global loads / stores in flight
  __device__ __forceinline__ void sync_lds() const { CAVE_LDS_WAIT(); }
  // LDS-only form of wave_fence (ds operations of one wave execute in order)
  __device__ __forceinline__ void wave_fence_lds() const { CAVE_WAVE_ORDER(); }
  // make this wave's earlier stores visible to its own later loads issued by other lanes
  __device__ __forceinline__ void wave_fence() const {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ double reduce_sum(double v) const { return wave_sum_f64(v); }
  __device__ __forceinline__ void reduce_sum2(double& a, double& b) const { a = wave_sum_f64(a); b = wave_sum_f64(b); }
  __device__ __forceinline__ double team_reduce_sum(double v) const { return quad_sum_f64(v); }
  __device__ __forceinline__ double reduce_max(double v) const { return wave_max_f64(v); }
  __device__ __forceinline__ uint32_t reduce_add_u32(uint32_t v) const { return wave_sum_u32(v); }
  __device__ __forceinline__ void atomic_add_u32(uint32_t* p, uint32_t v) const { atomicAdd(p, v); }
  __device__ __forceinline__ uint32_t atomic_inc_ret_u32(uint32_t* p) const { return atomicAdd(p, 1u); }
  __device__ __forceinline__ void atomic_or_u32(uint32_t* p, uint32_t v) const { atomicOr(p, v); }
  __device__ __forceinline__ void atomic_add_f64(double* p, double v) const { atomicAdd(p, v); }
  __device__ __forceinline__ void atomic_add_f64_lds(typename SpacePtr<double, 3>::type p, double v) const {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_f64
  }
  // 64-bit integer add in LDS (fixed-point accumulation: associative, so the sum does not depend on the order)
  __device__ __forceinline__ void atomic_add_i64_lds(typename SpacePtr<long long, 3>::type p, long long v) const {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u64
  }
  __device__ __forceinline__ void atomic_add_i64(long long* p, long long v) const {
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
  }

  // in-place exclusive scan of an LDS array; returns the total
  __device__ __forceinline__ uint32_t exclusive_scan_u32(uint32_t* a, int n) const {
    uint32_t carry = 0;
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      uint32_t v = (i < n) ? a[i] : 0u;
      uint32_t inc = wave_inclusive_scan_u32(v);
      if (i < n) a[i] = carry + inc - v;
      carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    __syncthreads();
    return carry;
  }
  // ordered list of i in [0,n) with (flags[i] & mask) == val
  __device__ __forceinline__ uint32_t compact_mask_u8(const uint8_t* flags, int n, uint8_t mask, uint8_t val,
                                                      uint32_t* out) const {
    uint32_t cnt = 0;
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      bool pr = (i < n) && ((flags[i] & mask) == val);
      uint64_t b = __ballot(pr);
      if (pr) out[cnt + mbcnt64(b)] = (uint32_t)i;
      cnt += (uint32_t)__popcll(b);
    }
    return cnt;
  }
  __device__ __forceinline__ uint32_t compact_nonzero_u8(const uint8_t* flags, int n, uint32_t* out) const {
    return compact_mask_u8(flags, n, 0xff, 0, out, true);
  }
  __device__ __forceinline__ uint32_t compact_mask_u8(const uint8_t* flags, int n, uint8_t mask, uint8_t val,
                                                      uint32_t* out, bool negate) const {
    uint32_t cnt = 0;
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      bool pr = (i < n) && (((flags[i] & mask) == val) != negate);
      uint64_t b = __ballot(pr);
      if (pr) out[cnt + mbcnt64(b)] = (uint32_t)i;
      cnt += (uint32_t)__popcll(b);
    }
    return cnt;
  }
  __device__ __forceinline__ void solve_spd(const double* H, int ldh, const double* g, const uint8_t* act, int p,
                                            double reg_rel, double* dv) const {
    gj_solve_small<64>(lane, H, ldh, g, act, p, reg_rel, dv);  // the low-register form: this kernel shape is built for two waves per SIMD
  }

  // Stream one dense instance (n floats, row-major) and append its non-zeros, in flat (row-major)
  // order, as (flat index, value) pairs to (eflat, eval).  Returns the number of non-zeros seen
  // (entries beyond `cap` are counted, not stored; eflat/eval need cap + NT slots, the last NT are
  // per-thread dump slots).  Row/column are derived afterwards, once per entry
  // (cone_instance.h scan_and_build), so the per-KiB loop stays short.
  // COND = true: eflat / eval live in global memory -> predicated stores instead of dump slots.
  template <bool COND = false>
  __device__ __forceinline__ uint32_t scan_dense(const float* __restrict__ A, uint32_t n, uint32_t* eflat, float* eval,
                                                 uint32_t cap) const {
    constexpr int U = SCAN_UNROLL;
    uint32_t cursor = 0;
    const uint32_t dump = cap + (uint32_t)lane;
    // head: elements before the first 16-byte boundary
    uint32_t head = (uint32_t)(((16u - (uint32_t)((uintptr_t)A & 15u)) & 15u) >> 2);
    if (head > n) head = n;
    if (head) scan_single(A, 0u, head, cursor, eflat, eval, cap);
    const float4* __restrict__ A4 = reinterpret_cast<const float4*>(A + head);
    const uint32_t n4 = (n - head) >> 2;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // main: batches of U x 1 KiB, software-pipelined two deep (the next batch's loads are in flight
    // while the current one is scanned), so HBM latency overlaps the ballot/emit work
    const uint32_t step = 64u * U;
    float4 bufA[U], bufB[U];
    auto load_batch = [&](float4* buf, uint32_t t) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // unconditional 16-byte load from a clamped index (a select around the load would be split
        // into four predicated dword loads); out-of-range chunks are zeroed when scanned
        uint32_t i = t + (uint32_t)u * 64u + (uint32_t)lane;
        buf[u] = A4[i < n4 ? i : n4 - 1u];
      }
    };
    auto scan_batch = [&](const float4* buf, uint32_t t) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t i = t + (uint32_t)u * 64u + (uint32_t)lane;
        float4 v = buf[u];
        if (i >= n4) v = z4;
        ChunkSlots s = chunk_slots(v);
        if (s.total == 0u) continue;  // wave-uniform
        if constexpr (COND) chunk_emit_cond(v, head + 4u * i, cursor, s.rel, s.nzm, eflat, eval, cap);
        else chunk_emit(v, head + 4u * i, cursor, s.rel, s.nzm, dump, eflat, eval, cap);
        cursor += s.total;
      }
    };
    if (n4 > 0) load_batch(bufA, 0);
    for (uint32_t t = 0; t < n4; t += 2u * step) {
      if (t + step < n4) load_batch(bufB, t + step);
      scan_batch(bufA, t);
      if (t + step < n4) {
        if (t + 2u * step < n4) load_batch(bufA, t + 2u * step);
        scan_batch(bufB, t + step);
      }
    }
    const uint32_t done = head + 4u * n4;
    if (done < n) scan_single(A, done, n, cursor, eflat, eval, cap);
    return cursor;
  }
  // fewer than 64 stray elements [lo, hi): one per lane
  __device__ __forceinline__ void scan_single(const float* __restrict__ A, uint32_t lo, uint32_t hi, uint32_t& cursor,
                                              uint32_t* eflat, float* eval, uint32_t cap) const {
    bool valid = lo + (uint32_t)lane < hi;
    float v = valid ? A[lo + lane] : 0.0f;
    bool nz = valid && (v != 0.0f);
    uint64_t m = __ballot(nz);
    uint32_t pos = cursor + mbcnt64(m);
    if (nz && pos < cap) { eflat[pos] = lo + (uint32_t)lane; eval[pos] = v; }
    cursor += (uint32_t)__popcll(m);
  }
};

}  // namespace cave
