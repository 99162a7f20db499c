// ctx_block.h — SPMD context: NW cooperating 64-lane waves per instance (workgroup = NW waves).
//
// Why: at the benchmark size (B = 1024 on 256 CUs) there is one instance per SIMD, and a lone wave
// is latency/issue-bound.  Giving each instance a 4-wave workgroup spreads its streaming scan, cone
// build and every vector phase of the Newton iteration over the CU's four SIMDs (waves of one
// workgroup are dealt to different SIMDs), while the register-resident Gauss-Jordan solve stays on
// wave 0.  LDS per instance is unchanged, so residency stays at 4 workgroups per CU.
//
// Cross-wave reductions: DPP inside each wave, one LDS slot per wave, ONE barrier (ping-pong
// buffers make the second barrier unnecessary), fixed summation order -> deterministic.
// Every method must be called by all threads of the workgroup from uniform control flow.
#pragma once
#include "wave_prims.h"
#include "ctx_wave.h"

namespace cave {

// SoloCtx — the Newton iteration of a SMALL +-1 cone on ONE wave (wave 0 of a multi-wave workgroup, the other
// waves parked at the workgroup barrier that follows; or the only wave of a one-wave workgroup), over the
// "lite" index structures of cone_core.h.  A cross-lane reduction is a DPP tree with no LDS slot and no
// barrier, and a phase boundary is free: the LDS operations of one wave execute in issue order, so only the
// compiler has to be kept from moving memory operations across it.
template <int PMAX_, int KREG_>
struct SoloCtx {
  static constexpr int NT = 64;
  static constexpr int TEAM = 4;
  static constexpr int PMAX = PMAX_;
  static constexpr int KREG = KREG_;
  static constexpr int NWAVES = 1;
  static constexpr int WL = 64;
  static constexpr bool LITE = true;
  int lane;
  LiteCone lite;
#ifdef CAVE_STAMPS
  unsigned long long* st;
#endif
  __device__ __forceinline__ int tid() const { return lane; }
  __device__ __forceinline__ int wave_id() const { return 0; }
  __device__ __forceinline__ int lane_id() const { return lane; }
  __device__ __forceinline__ double wave_sum(double v) const { return wave_sum_f64(v); }
  __device__ __forceinline__ double wave_max(double v) const { return wave_max_f64(v); }
  __device__ __forceinline__ void sync() const { CAVE_WAVE_ORDER(); }
  __device__ __forceinline__ double reduce_sum(double v) const { return wave_sum_f64(v); }
  __device__ __forceinline__ void reduce_sum2(double& a, double& b) const { wave_reduce2_f64<false>(a, b); }
  __device__ __forceinline__ void reduce_sum_max(double& s, double& m) const { wave_reduce2_f64<true>(s, m); }
  __device__ __forceinline__ double reduce_max(double v) const { return wave_max_f64(v); }
  __device__ __forceinline__ uint32_t reduce_add_u32(uint32_t v) const { return wave_sum_u32(v); }
  __device__ __forceinline__ double team_reduce_sum(double v) const { return quad_sum_f64(v); }
  __device__ __forceinline__ void atomic_add_f64(double* p, double v) const { atomicAdd(p, v); }
  // the Schur system of the bound rows (lite_model_step, cone_core.h): at most 8 rows
  __device__ __forceinline__ void solve_spd(const double* H, int ldh, const double* g, const uint8_t* act, int p,
                                            double reg_rel, double* dv) const {
    gj_solve_regs<8, true>(lane, H, ldh, g, act, p, reg_rel, dv);
  }
};

// WIDE: launched with a 256-VGPR budget even at 4 waves (the large-cone kernels): deeper scan prefetch
template <int NW, bool WIDE = false>
struct BlockCtx {
  static constexpr int NT = 64 * NW;
  static constexpr int TEAM = 4;
  static constexpr int SCAN_UNROLL = (NW <= 2 || WIDE) ? 8 : 4;  // KiB per wave per batch (two batches in flight); VGPR budget
  static constexpr int PMAX = (NW <= 2 || WIDE) ? 64 : 32;  // register budget: 128 VGPRs at 4 waves/SIMD, 256 at 2 (or WIDE)
  static constexpr int MIN_WAVES_PER_EU = WIDE ? 2 : (NW <= 2 ? NW : 4);  // (WIDE: 256 registers -- two such workgroups share a compute unit in the diet tier; at 1 the w8 packed kernel took 305 and only one fitted)
  static constexpr int KREG = 2;         // line search keeps r, q in registers when d <= KREG * NT
  static constexpr bool LITE_OK = (NW == 2 && !WIDE);  // 256-register budget at full residency (two waves per SIMD)
  // carries the "diet" LDS layout of the packed operator (H as a packed triangle, CSC + avg read in place from the
  // store): the four-wave wide shape, which is what cones of the TSP-50 class (80 - 160 KB of LDS) are launched with
  static constexpr bool DIET_OK = (NW == 4 && WIDE);
  struct Scratch {
    double f64[2][8];
    uint32_t u32[2][8];
  };
  static constexpr uint32_t SCRATCH_BYTES = 256;
  static constexpr int NWAVES = NW;
  static constexpr int WL = 64;
  static_assert(sizeof(Scratch) <= SCRATCH_BYTES && NW <= 8, "scratch layout");
  int lane, wave, t;
  uint32_t par;
  typename SpacePtr<Scratch, 3>::type sc;  // LDS-typed: ds_* also inside real calls (a generic pointer member compiles to flat_*)
#ifdef CAVE_STAMPS
  unsigned long long st[32];  // [0,16) exported per instance; [16,32) scratch slots of the fine stamps
#endif
  __device__ __forceinline__ void init(unsigned char* smem) {
    t = (int)threadIdx.x;
    lane = t & 63;
    wave = t >> 6;
    par = 0;
    sc = space_cast<3>(reinterpret_cast<Scratch*>(smem));
  }
  // hand three words from wave 0 to the whole workgroup (one barrier)
  __device__ __forceinline__ void broadcast_from_wave0(double& a, int& b, int& c) {
    if (t == 0) { sc->f64[par][0] = a; sc->u32[par][0] = (uint32_t)b; sc->u32[par][1] = (uint32_t)c; }
    __syncthreads();
    a = sc->f64[par][0]; b = (int)sc->u32[par][0]; c = (int)sc->u32[par][1];
    par ^= 1u;
  }
  __device__ __forceinline__ int tid() const { return t; }
  __device__ __forceinline__ int wave_id() const { return wave; }
  __device__ __forceinline__ int lane_id() const { return lane; }
  // sum / max over the lanes of the calling wave only (no barrier; every lane gets the result)
  __device__ __forceinline__ double wave_sum(double v) const { return wave_sum_f64(v); }
  __device__ __forceinline__ double wave_max(double v) const { return wave_max_f64(v); }
  // lane l gets lane l-1's value (lane 0: 0)
  __device__ __forceinline__ double wave_shift_up(double v) const { return dpp_f64<0x138, 0xf>(0.0, v); }  // wave_shr:1
  // barrier that orders LDS traffic only (typed ds_* accesses): does not wait for global loads / stores in flight
  __device__ __forceinline__ void sync_lds() const { CAVE_LDS_BARRIER(); }
  // LDS-only form of wave_fence (ds operations of one wave execute in order)
  __device__ __forceinline__ void wave_fence_lds() const { CAVE_WAVE_ORDER(); }
  // make this wave's earlier stores visible to its own later loads issued by other lanes
  __device__ __forceinline__ void wave_fence() const {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void sync() const { __syncthreads(); }

  __device__ __forceinline__ double reduce_sum(double v) {
    double w = wave_sum_f64(v);
    if (lane == 0) sc->f64[par][wave] = w;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += sc->f64[par][i];
    par ^= 1u;
    return s;
  }
  // two sums for the price of one barrier (needs NW <= 4: slots wave and 4 + wave)
  __device__ __forceinline__ void reduce_sum2(double& a, double& b) {
    static_assert(NW <= 4, "reduce_sum2 uses 2*NW scratch slots");
    double wa = wave_sum_f64(a), wb = wave_sum_f64(b);
    if (lane == 0) { sc->f64[par][wave] = wa; sc->f64[par][4 + wave] = wb; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { sa += sc->f64[par][i]; sb += sc->f64[par][4 + i]; }
    par ^= 1u;
    a = sa; b = sb;
  }
  __device__ __forceinline__ double reduce_max(double v) {
    double w = wave_max_f64(v);
    if (lane == 0) sc->f64[par][wave] = w;
    __syncthreads();
    double s = sc->f64[par][0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s = fmax(s, sc->f64[par][i]);
    par ^= 1u;
    return s;
  }
  __device__ __forceinline__ uint32_t reduce_add_u32(uint32_t v) {
    uint32_t w = wave_sum_u32(v);
    if (lane == 0) sc->u32[par][wave] = w;
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += sc->u32[par][i];
    par ^= 1u;
    return s;
  }
  __device__ __forceinline__ double team_reduce_sum(double v) const { return quad_sum_f64(v); }
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

  // publish this wave's count, return (sum over lower waves, total)
  // (the exchange goes through LDS only, so the barrier waits for LDS only: global loads a caller has in
  // flight -- the scan's prefetched batch -- stay in flight across it; __syncthreads() would drain them)
  __device__ __forceinline__ void wave_prefix(uint32_t wcount, uint32_t& pre, uint32_t& tot) {
    if (lane == 0) sc->u32[par][wave] = wcount;
    sync_lds();
    pre = 0;
    tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      uint32_t x = sc->u32[par][i];
      if (i < wave) pre += x;
      tot += x;
    }
    par ^= 1u;
  }

  __device__ __forceinline__ uint32_t exclusive_scan_u32(uint32_t* a, int n) {
    uint32_t carry = 0;
    for (int base = 0; base < n; base += NT) {
      int i = base + t;
      uint32_t v = (i < n) ? a[i] : 0u;
      uint32_t inc = wave_inclusive_scan_u32(v);
      uint32_t pre, tot;
      wave_prefix((uint32_t)__builtin_amdgcn_readlane((int)inc, 63), pre, tot);
      if (i < n) a[i] = carry + pre + inc - v;
      carry += tot;
    }
    __syncthreads();
    return carry;
  }
  __device__ __forceinline__ uint32_t compact_mask_u8(const uint8_t* flags, int n, uint8_t mask, uint8_t val,
                                                      uint32_t* out, bool negate = false) {
    uint32_t cnt = 0;
    for (int base = 0; base < n; base += NT) {
      int i = base + t;
      bool pr = (i < n) && (((flags[i] & mask) == val) != negate);
      uint64_t b = __ballot(pr);
      uint32_t pre, tot;
      wave_prefix((uint32_t)__popcll(b), pre, tot);
      if (pr) out[cnt + pre + mbcnt64(b)] = (uint32_t)i;
      cnt += tot;
    }
    return cnt;
  }
  __device__ __forceinline__ uint32_t compact_nonzero_u8(const uint8_t* flags, int n, uint32_t* out) {
    return compact_mask_u8(flags, n, 0xff, 0, out, true);
  }

  // wave 0 solves in registers; the caller's sync() publishes dv to the other waves
  __device__ __forceinline__ void solve_spd(const double* H, int ldh, const double* g, const uint8_t* act, int p,
                                            double reg_rel, double* dv) const {
    if (wave == 0) {
      if constexpr (WIDE && NW == 4) gj_solve_wide_call<false>(lane, H, ldh, g, act, p, reg_rel, dv);
      else gj_solve_small<PMAX>(lane, H, ldh, g, act, p, reg_rel, dv);
    }
  }
  // H as the packed lower triangle (diet layout).  (Tried, round 4: the two workgroups that share a compute unit in
  // that tier solving on DIFFERENT SIMDs -- wave 0 or wave 2 by the parity of the hardware wave slot: TSP-50, B = 512
  // 0.354 ms against 0.336 ms with both on their wave 0.  Dropped.)
  __device__ __forceinline__ void solve_spd_tri(const double* H, const double* g, const uint8_t* act, int p, double reg_rel,
                                                double* dv) const {
    if (wave == 0) {
      if constexpr (WIDE && NW == 4) gj_solve_wide_call<true>(lane, H, 0, g, act, p, reg_rel, dv);
      else gj_solve_small<PMAX, true>(lane, H, 0, g, act, p, reg_rel, dv);
    }
  }

  // Cooperative ordered streaming scan (contract: see WaveCtx::scan_dense).  Per round, wave w owns
  // the U consecutive 1 KiB chunks [w*U, (w+1)*U) of a NW*U KiB window, so flat order is
  // wave-major; one barrier per round turns the per-wave non-zero counts into slot bases.
  // U: KiB per wave per batch (two batches in flight)
  template <bool COND = false, int U = SCAN_UNROLL>
  __device__ __forceinline__ uint32_t scan_dense(const float* __restrict__ A, uint32_t n, uint32_t* eflat, float* eval,
                                                 uint32_t cap) {
    static_assert(U <= 16, "component masks of a batch are packed into 64 bits");
    uint32_t cursor = 0;
    const uint32_t dump = cap + (uint32_t)t;
    uint32_t head = (uint32_t)(((16u - (uint32_t)((uintptr_t)A & 15u)) & 15u) >> 2);
    if (head > n) head = n;
    if (head) scan_single(A, 0u, head, cursor, eflat, eval, cap);
    const float4* __restrict__ A4 = reinterpret_cast<const float4*>(A + head);
    const uint32_t n4 = (n - head) >> 2;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t round = 64u * U * NW;
    const uint32_t woff = (uint32_t)wave * 64u * U + (uint32_t)lane;
    float4 bufA[U], bufB[U];
    auto load_batch = [&](float4* buf, uint32_t r0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t i = r0 + woff + (uint32_t)u * 64u;
        buf[u] = CAVE_NT_LOAD_F4(&A4[i < n4 ? i : n4 - 1u]);  // unconditional dwordx4 from a clamped index; read once: nt
      }
    };
    auto scan_batch = [&](const float4* buf, uint32_t r0) {
      uint32_t rel[U], wsum = 0;
      uint64_t nzm = 0, cmv = 0;  // 4 bits per chunk: this lane's component mask / the wave's (uniform)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t i = r0 + woff + (uint32_t)u * 64u;
        float4 v = buf[u];
        if (i >= n4) v = z4;
        ChunkSlots s = chunk_slots(v);
        rel[u] = wsum + s.rel;
        nzm |= (uint64_t)s.nzm << (4 * u);
        wsum += s.total;
        cmv |= (uint64_t)s.cm << (4 * u);
      }
      uint32_t pre, tot;
      wave_prefix(wsum, pre, tot);
      if (tot != 0u) {  // workgroup-uniform
        const uint32_t base = cursor + pre;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t cm = (uint32_t)(cmv >> (4 * u)) & 15u;
          if (cm == 0u) continue;  // this wave's chunk u holds no non-zero at all
          uint32_t i = r0 + woff + (uint32_t)u * 64u;
          const uint32_t nz4 = (uint32_t)(nzm >> (4 * u)) & 15u;
          if constexpr (COND) chunk_emit_cond(buf[u], head + 4u * i, base, rel[u], nz4, eflat, eval, cap);
          else chunk_emit(buf[u], head + 4u * i, base, rel[u], nz4, dump, eflat, eval, cap);
        }
      }
      cursor += tot;
    };
    if (n4 > 0) load_batch(bufA, 0);
    for (uint32_t r0 = 0; r0 < n4; r0 += 2u * round) {
      if (r0 + round < n4) load_batch(bufB, r0 + round);
      scan_batch(bufA, r0);
      if (r0 + round < n4) {
        if (r0 + 2u * round < n4) load_batch(bufA, r0 + 2u * round);
        scan_batch(bufB, r0 + round);
      }
    }
    const uint32_t done = head + 4u * n4;
    if (done < n) scan_single(A, done, n, cursor, eflat, eval, cap);
    return cursor;
  }
  // The scan of the step kernel's pack half (256-register budget), same contract and output order as scan_dense but
  // PREDICATED stores (no dump slots: eflat / eval hold `cap` entries), for SPARSE cones -- what this domain has: a
  // TSP-20 instance holds ~1 100 non-zeros in 44 080 elements, one or two per KiB.  scan_dense runs ~100 instructions
  // per KiB chunk whatever it holds; here a chunk costs ~8 when it is empty (one OR-test + ballot), ~45 when no lane
  // holds more than one non-zero (ONE ballot + mbcnt gives the slots; the lanes with a non-zero store it under their
  // exec mask), the full four-ballot form only when some lane's float4 holds two or more (runs of consecutive edges
  // in TSP degree / cut rows).  U = 8 KiB per wave per batch: deeper batches (16: 32 KiB in flight per wave) shorten a
  // LONE workgroup's scan (tools/micro/stream_wg.hip: ~5 k cycles of memory latency per round trip) but not the
  // step kernel's, which runs two to four pack workgroups per compute unit beside the solve waves and is bound by
  // issue and instruction fetch (SQ_WAIT_INST_ANY 20 % of its wave cycles): U = 4 / 8 / 16 -> pack-only launch 66.4 /
  // 66.5 / 71.9 us, fused step 133.5 / 133.2 / 135.7 us.
  // Tried and dropped (round 4, same timing within 5 %): dword loads with one element per lane and a wave's whole
  // share resident in registers (one ballot per 64 floats, one barrier per round, buffer-load range checks).
  template <int U>
  __device__ __forceinline__ uint32_t scan_dense_sparse(const float* __restrict__ A, uint32_t n, uint32_t* eflat, float* eval,
                                                        uint32_t cap) {
    static_assert(U <= 16, "component masks of a batch are packed into 64 bits");
    uint32_t cursor = 0;
    uint32_t head = (uint32_t)(((16u - (uint32_t)((uintptr_t)A & 15u)) & 15u) >> 2);
    if (head > n) head = n;
    if (head) scan_single(A, 0u, head, cursor, eflat, eval, cap);
    const float4* __restrict__ A4 = reinterpret_cast<const float4*>(A + head);
    const uint32_t n4 = (n - head) >> 2;
    const uint32_t round = 64u * U * NW;
    const uint32_t woff = (uint32_t)wave * 64u * U + (uint32_t)lane;
    float4 bufA[U], bufB[U];
    auto load_batch = [&](float4* buf, uint32_t r0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t i = r0 + woff + (uint32_t)u * 64u;
        buf[u] = CAVE_NT_LOAD_F4(&A4[i < n4 ? i : n4 - 1u]);  // unconditional dwordx4 from a clamped index; read once: nt
      }
    };
    auto scan_batch = [&](const float4* buf, uint32_t r0) {
      uint32_t rel[U], wsum = 0;
      uint64_t nzm = 0;            // 4 bits per chunk: this lane's component mask
      uint32_t anym = 0, multim = 0;  // wave-uniform, one bit per chunk: not empty / some lane holds two or more
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t i = r0 + woff + (uint32_t)u * 64u;
        const float4 v = buf[u];
        // any non-zero in this lane's float4?  (+-0 have no bits below the sign; NaN counts, as `x != 0.0f` does)
        const uint32_t bits = (f2u(v.x) | f2u(v.y) | f2u(v.z) | f2u(v.w)) & 0x7fffffffu;
        const bool mine = bits != 0u && i < n4;
        const uint64_t any = __ballot(mine);
        rel[u] = wsum;
        if (any == 0ull) continue;  // wave-uniform: the chunk is empty
        anym |= 1u << u;
        const uint32_t n0 = mine && v.x != 0.0f, n1 = mine && v.y != 0.0f, n2 = mine && v.z != 0.0f, n3 = mine && v.w != 0.0f;
        const uint32_t m4 = n0 | (n1 << 1) | (n2 << 2) | (n3 << 3);
        nzm |= (uint64_t)m4 << (4 * u);
        const uint64_t multi = __ballot((m4 & (m4 - 1u)) != 0u);
        if (multi == 0ull) {  // wave-uniform: one non-zero per lane at most
          rel[u] += mbcnt64(any);
          wsum += (uint32_t)__popcll(any);
        } else {
          multim |= 1u << u;
          const uint64_t m0 = __ballot(n0 != 0u), m1 = __ballot(n1 != 0u), m2 = __ballot(n2 != 0u), m3 = __ballot(n3 != 0u);
          rel[u] += mbcnt64(m0) + mbcnt64(m1) + mbcnt64(m2) + mbcnt64(m3);
          wsum += (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
        }
      }
      uint32_t pre, tot;
      wave_prefix(wsum, pre, tot);
      if (tot != 0u) {  // workgroup-uniform
        const uint32_t base = cursor + pre;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!((anym >> u) & 1u)) continue;  // wave-uniform
          const uint32_t i = r0 + woff + (uint32_t)u * 64u;
          const uint32_t m4 = (uint32_t)(nzm >> (4 * u)) & 15u;
          const uint32_t f = head + 4u * i, p0 = base + rel[u];
          const float4 v = buf[u];
          if (!((multim >> u) & 1u)) {
            if (m4 != 0u && p0 < cap) {  // the lanes holding the chunk's non-zeros (one each)
              const uint32_t comp = (uint32_t)__builtin_ctz(m4);
              eflat[p0] = f + comp;
              eval[p0] = (m4 & 1u) ? v.x : (m4 & 2u) ? v.y : (m4 & 4u) ? v.z : v.w;
            }
          } else {
            const uint32_t p1 = p0 + (m4 & 1u), p2 = p1 + ((m4 >> 1) & 1u), p3 = p2 + ((m4 >> 2) & 1u);
            if ((m4 & 1u) && p0 < cap) { eflat[p0] = f; eval[p0] = v.x; }
            if ((m4 & 2u) && p1 < cap) { eflat[p1] = f + 1u; eval[p1] = v.y; }
            if ((m4 & 4u) && p2 < cap) { eflat[p2] = f + 2u; eval[p2] = v.z; }
            if ((m4 & 8u) && p3 < cap) { eflat[p3] = f + 3u; eval[p3] = v.w; }
          }
        }
      }
      cursor += tot;
    };
    if (n4 > 0) load_batch(bufA, 0);
    for (uint32_t r0 = 0; r0 < n4; r0 += 2u * round) {
      if (r0 + round < n4) load_batch(bufB, r0 + round);
      scan_batch(bufA, r0);
      if (r0 + round < n4) {
        if (r0 + 2u * round < n4) load_batch(bufA, r0 + 2u * round);
        scan_batch(bufB, r0 + round);
      }
    }
    const uint32_t done = head + 4u * n4;
    if (done < n) scan_single(A, done, n, cursor, eflat, eval, cap);
    return cursor;
  }
  // fewer than 64 stray elements [lo, hi): wave 0, one per lane
  __device__ __forceinline__ void scan_single(const float* __restrict__ A, uint32_t lo, uint32_t hi, uint32_t& cursor,
                                              uint32_t* eflat, float* eval, uint32_t cap) {
    bool valid = (wave == 0) && (lo + (uint32_t)lane < hi);
    float v = valid ? A[lo + lane] : 0.0f;
    bool nz = valid && (v != 0.0f);
    uint64_t m = __ballot(nz);
    uint32_t pre, tot;
    wave_prefix((uint32_t)__popcll(m), pre, tot);
    uint32_t pos = cursor + pre + mbcnt64(m);
    if (nz && pos < cap) { eflat[pos] = lo + (uint32_t)lane; eval[pos] = v; }
    cursor += tot;
  }
};

}  // namespace cave
