// cone_instance.h — what one workgroup does for one instance, written against a Ctx.
// Shared by the HIP kernels (cave_hip.hip, Ctx = WaveCtx) and the serial test
// build (tests/emul, Ctx = SerialCtx).
#pragma once
#include "../../include/cave_hip.h"
#include "cone_common.h"
#include "cone_core.h"

namespace cave {

CAVE_HOSTDEV int32_t packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz, bool all_pm1, bool lite_room, bool diet);

struct OutPtrs {
  float* proj;
  float* rnorm;
  float* target;
  float* loss;
  float* grad;
  int32_t* status;
  int32_t* iters;
};

struct DenseParams {
  const float* ctrs;
  const float* pred;
  int64_t B;
  int32_t m, d;
  int32_t mode;
  float sign, inner_ratio;
  int32_t max_iter;
  uint32_t nnz_cap, lds_bytes;
  OutPtrs o;
};

struct PackParams {
  const float* ctrs;
  int64_t B;
  int32_t m, d;
  uint32_t nnz_cap, lds_bytes;
  int32_t* n_rows;
  int32_t* n_nnz;
  int32_t* status;
  cave_cone_store store;  // fill pass only
  int64_t slot0;
  int32_t fill;
};

struct PackedParams {
  cave_cone_store store;
  const int64_t* ids;
  const float* pred;
  int64_t B;
  int32_t mode;
  float sign, inner_ratio;
  int32_t max_iter;
  uint32_t lds_bytes;
  OutPtrs o;
};

CAVE_HD float quiet_nan() {
  union { uint32_t u; float f; } x;
  x.u = 0x7fc00000u;
  return x.f;
}

template <class C>
CAVE_HD void fill_failure(C& c, int d, int64_t b, const OutPtrs& o) {
  const float nanv = quiet_nan();
  for (int k = c.tid(); k < d; k += C::NT) {
    if (o.proj) o.proj[b * d + k] = nanv;
    if (o.target) o.target[b * d + k] = nanv;
    if (o.grad) o.grad[b * d + k] = nanv;
  }
  if (c.tid() == 0) {
    if (o.rnorm) o.rnorm[b] = nanv;
    if (o.loss) o.loss[b] = nanv;
  }
}

// scan the dense block + build the reduced cone; returns status
// (LARGE: the arena is global memory -> predicated scan stores, no dump slots)
// DEEP (multi-wave contexts with the 256-register budget: the step kernel's pack half): BlockCtx::scan_dense_sparse,
// predicated stores -- no dump slots behind the scan output
template <class C, bool LARGE = false, bool DEEP = false>
CAVE_HD int32_t scan_and_build(C& c, Arena& ar, ConeBuild& cb, const float* A, int m, int d, uint32_t cap) {
  cb.d = d;
  cb.m = m;
  // (row << 16) | col packing of the scan output, 32-bit flat index
  if (m > 0xffff || d > 0xffff || (uint64_t)m * (uint64_t)d > 0xffffffffull) return ST_TOO_LARGE;
  // scan output = build-phase temporaries at the top of the arena (+ per-thread dump slots)
  const uint32_t dump_slots = (LARGE || DEEP) ? 0u : (uint32_t)C::NT;
  cb.erc = ar.get_top<uint32_t>(cap + dump_slots);
  cb.eall = ar.get_top<float>(cap + dump_slots);
  cb.rptr = ar.get_top<uint32_t>((uint32_t)m + 1u);
  if (ar.ovf) return ST_TOO_LARGE;
  for (int r = c.tid(); r <= m; r += C::NT) cb.rptr[r] = 0u;
  c.sync();
  CAVE_T0();
  uint32_t nnz;  // (erc = flat index for now)
  if constexpr (DEEP) nnz = c.template scan_dense_sparse<8>(A, (uint32_t)m * (uint32_t)d, cb.erc, cb.eall, cap);
  else nnz = c.template scan_dense<LARGE>(A, (uint32_t)m * (uint32_t)d, cb.erc, cb.eall, cap);
  c.sync();
  CAVE_ACC(10);
  if (nnz > cap) return ST_TOO_LARGE;
  // flat index f = row*d + col  ->  (row << 16) | col, and the per-row counts
  const double inv_d = 1.0 / (double)d;
  for (uint32_t e = c.tid(); e < nnz; e += C::NT) {
    const uint32_t f = cb.erc[e];
    uint32_t row = (uint32_t)((double)f * inv_d);  // within +-1 of f / d for f < 2^32
    int32_t col = (int32_t)(f - row * (uint32_t)d);
    if (col < 0) { row -= 1u; col += d; }
    else if (col >= d) { row += 1u; col -= d; }
    cb.erc[e] = (row << 16) | (uint32_t)col;
    c.atomic_add_u32(&cb.rptr[row], 1u);
  }
  c.sync();
  cb.nnz_all = nnz;
  return build_cone(c, ar, cb);
}

CAVE_HD SolveView view_of(const ConeBuild& cb) {
  SolveView v;
  v.d = cb.d; v.p = cb.p; v.n_valid = cb.n_valid_proj; v.pm1 = cb.pm1;
  v.mptr = cb.mptr; v.mcol = cb.mcol; v.mval = cb.mval; v.vkind = cb.vkind;
  v.cptr = cb.cptr; v.cvar = cb.cvar; v.cvalc = cb.cvalc; v.usign = cb.usign;
  v.nlong = 0; v.longrow = nullptr;
  return v;
}

// hot-first allocation of the large-cone path: LDS while it lasts, then the global workspace
#ifdef CAVE_EMUL_COUNTERS
inline long* emul_counters() {  // test builds only: [0] dense path, [1] one-wave band path, [2] lite path, [3] H-free band, [4] lite with bound rows, [5] red-black reduction of the band
  static long cnt[8] = {0};
  return cnt;
}
#endif

template <class T>
CAVE_HD T* hot_get(Arena* hot, Arena& ar, uint32_t n) {
  if (hot) {
    T* q = hot->try_get<T>(n);
    if (q) return q;
  }
  return ar.get<T>(n);
}

// Solve + epilogue for one instance whose SolveView is ready.  y must be loaded.
// LARGE: band Hessian + solve_spd_band, work arrays hot-first (`hot` = LDS arena, `ar` = global workspace).
// warm_theta / warm_state: this instance's slice of the store's multiplier cache (global memory) and its
// state byte, or null (no warm start): read as the starting point when the state is 1, rewritten after a
// converged solve, invalidated after a failed one.
template <class C, bool LARGE = false>
CAVE_HD int32_t solve_and_finish(C& c, Arena& ar, Arena* hot, const SolveView& v, int mode, float sign,
                                 float inner_ratio, int max_iter, float* y, const float* avg, int64_t b,
                                 const OutPtrs& o, int* iters_out, float* warm_theta = nullptr,
                                 uint8_t* warm_state = nullptr, unsigned char* rb_cache = nullptr, uint64_t rb_cache_bytes = 0) {
  const int d = v.d;
  const bool need_proj = (mode == MODE_PROJECT || mode == MODE_EXACT || mode == MODE_INNER || mode == MODE_IPM);
  CAVE_T0();  // (stamp builds, large path: slot 0 = set-up before the solver, 1 = the solver call, 9 = epilogue)
  int32_t st = ST_OK;
  double f = 0.0;
  const bool empty = (v.n_valid == 0);
  *iters_out = 0;
  double* res = nullptr;
  double* tvec = nullptr;
  if constexpr (!LARGE) {
    res = ar.get<double>(d);
    tvec = ar.get<double>(d);
    if (ar.ovf) return ST_TOO_LARGE;
  }
  if (need_proj && !empty) {
    const int p = v.p;
    const uint32_t pp = (uint32_t)(p > 0 ? p : 1);
    SolveWork w;
    w.y = y;
    w.ls_on = false;
    w.bw = 0; w.band_wave = false; w.bwin = nullptr; w.bfac = nullptr; w.bz = nullptr; w.bstg = nullptr; w.bch = 0; w.band_hot = false;
    if (mode == MODE_IPM) { warm_theta = nullptr; warm_state = nullptr; }  // the interior iterate is not a starting point
    // ONE thread reads the state byte and the workgroup agrees on it through a reduction: another workgroup
    // solving the same store slot (a repeated id in the batch) may be rewriting it right now, and a decision that
    // differed between waves would send them into different barrier sequences below.
    bool warm_on = false;
    if (warm_theta && warm_state) warm_on = c.reduce_add_u32(c.tid() == 0 ? (uint32_t)(*warm_state == 1) : 0u) != 0u;
    w.warm = warm_on ? warm_theta : nullptr;
    bool rb_wanted = false;
    (void)rb_wanted;
    double gcol_bound = 0.0;  // (longest row) x (largest |entry|) of the reduced cone: scale of the column-wise gradient
    (void)gcol_bound;
    if constexpr (LARGE) {
      const int bw = band_halfwidth(c, v);
      const uint32_t ld = (uint32_t)bw + 1u;
      if ((uint64_t)pp * ld > (1ull << 27) || ld > 4096u) return ST_TOO_LARGE;  // 1 GiB of band per instance
      w.bw = bw;
      w.ldh = (int)ld;
      // Dense (or widely banded) reduced systems of up to 128 rows -- TSP-100 -- keep the whole matrix in LDS,
      // folded, and factor it once per Newton iteration (cone_dense.h).
      w.dn.on = false;
      w.gen.on = false;
      w.rb.on = false;
      {  // fixed-point scale of the Hessian accumulation: |H_ab| <= (largest entry)^2 * (longest row)
        double vm = v.pm1 ? 1.0 : 0.0, ml = 1.0;
        if (!v.pm1) {  // (eight entries per thread in flight: the values are in global memory on this path)
          const uint32_t nz = v.mptr[p];
          for (uint32_t e0 = c.tid(); e0 < nz; e0 += 8u * (uint32_t)C::NT) {
            float mv[8];
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) mv[j] = v.mval[e0 + j * (uint32_t)C::NT < nz ? e0 + j * (uint32_t)C::NT : nz - 1u];
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) vm = fmax(vm, fabs((double)mv[j]));
          }
        }
        for (int i = c.tid(); i < p; i += C::NT) ml = fmax(ml, (double)(v.mptr[i + 1] - v.mptr[i]));
        vm = c.reduce_max(vm);
        ml = c.reduce_max(ml);
        w.hscale = fixed_scale(vm, vm * ml);
        w.hinv = 1.0 / w.hscale;
        gcol_bound = vm * ml;
      }
      if (hot && p >= 1 && dense_shape(p, bw)) {
        uint32_t cnt = 0;
        for (int i = c.tid(); i < p; i += C::NT) cnt += v.vkind[i] ? 0u : 1u;
        const int nI = (int)c.reduce_add_u32(cnt);
        if (nI <= kDenseMaxBound && nI <= C::PMAX && (uint64_t)(hot->top - hot->off) >= dense_lds_bytes(p, nI) + 64u) {
          DenseWork& dn = w.dn;
          const uint32_t ldS = (uint32_t)(nI | 1), nb = (uint32_t)(nI > 0 ? nI : 1);
          dn.A = hot->try_get<double, 16u>(fold_entries(p));
          dn.dinv = hot->try_get<double>(pp);
          dn.z = hot->try_get<double>(pp);
          dn.x = hot->try_get<double>(pp);
          dn.scr = hot->try_get<double, 16u>(dense_scratch_entries(p));
          dn.S = hot->try_get<double>(nb * ldS);
          dn.sg = hot->try_get<double>(nb);
          dn.st = hot->try_get<double>(nb);
          dn.ss = hot->try_get<double>(nb);
          dn.sr = hot->try_get<double>(nb);
          dn.pos = hot->try_get<uint16_t>(pp);
          dn.ord = hot->try_get<uint16_t>(pp);
          dn.sact = hot->try_get<uint8_t>(nb);
          uint32_t* tmp = ar.get<uint32_t>(pp);
          if (ar.ovf) return ST_TOO_LARGE;
          if (dn.A && dn.dinv && dn.z && dn.x && dn.scr && dn.S && dn.sg && dn.st && dn.ss && dn.sr && dn.pos && dn.ord && dn.sact) {
            dn.on = true;
            dense_order(c, v, dn, tmp);
            dn.hscale = w.hscale;
            dn.hinv = w.hinv;
#ifdef CAVE_EMUL_COUNTERS
            if (c.tid() == 0) ++emul_counters()[0];  // test builds: how many instances took the dense path
#endif
          }
        }
      }
      // narrow bands: the whole elimination runs on one wave (cone_band.h, solve_spd_band_wave)
      bool wave_mode = false;
#if defined(CAVE_GPU_CODE) && !defined(CAVE_NO_BAND_WAVE)  // (diagnostic builds can pin the team form of the band solver)
      if constexpr (C::WL == 64) {
        const uint64_t need = 8ull * (band_wave_region(bw, p) + 2ull * pp) + pp + 64u;
        wave_mode = hot != nullptr && band_wave_fits(bw, p) && (uint64_t)(hot->top - hot->off) >= need;
      }
#endif
      // small, touched every elimination step / every inner round: LDS first
      if (w.dn.on) wave_mode = false;
      if (!w.dn.on) {
      w.bwin = hot_get<double>(hot, ar, wave_mode ? band_wave_region(bw, p) : ld * ld);  // (wave form: ring + staging in one piece)
      w.bz = hot_get<double>(hot, ar, pp);
      }
      w.step = hot_get<double>(hot, ar, pp);
      w.act = hot_get<uint8_t>(hot, ar, pp);
      // staging chunks: as many rows as the prefetch registers hold, fewer if that keeps them in LDS
      w.bch = band_chunk_rows<C>((int)ld);
      if (wave_mode) w.bch = (int)ld;
      else if (hot) {
        const uint32_t room = (hot->top - hot->off) / (2u * 8u * ld);
        if (room >= 4u && room < (uint32_t)w.bch) w.bch = (int)room;
      }
      if (!w.dn.on) {
#if defined(CAVE_GPU_CODE)
      if (wave_mode) w.bstg = w.bwin;  // (not used by the wave form)
      else
#endif
      w.bstg = hot_get<double>(hot, ar, 2u * (uint32_t)w.bch * ld);
      }
      w.g2 = hot_get<double>(hot, ar, pp);
      w.ttry = hot_get<double>(hot, ar, pp);
      w.told = hot_get<double>(hot, ar, pp);
      w.dv = hot_get<double>(hot, ar, pp);
      w.theta = hot_get<double>(hot, ar, pp);
      w.g = hot_get<double>(hot, ar, pp);
      w.rc = hot_get<double>(hot, ar, d);
      res = hot_get<double>(hot, ar, d);
      tvec = hot_get<double>(hot, ar, d);
      w.wold = nullptr;
      if (!w.dn.on) {
      w.H = ar.get<double>(pp * ld);
      w.bfac = ar.get<double>(pp * ld);
#if defined(CAVE_GPU_CODE)
      w.band_hot = hot && hot->owns(w.bwin) && hot->owns(w.bz) && hot->owns(w.step) && hot->owns(w.act) &&
                   hot->owns(w.bstg);
      w.band_wave = wave_mode && w.band_hot;
      // failure word of the two-wave hand-over (cone_band.h kBandSpinLimit): cleared here, examined after the solver
      if (w.band_wave && c.tid() == 0) reinterpret_cast<int*>(w.bwin + band_wave_flags_at(bw))[4] = 0;
#ifdef CAVE_EMUL_COUNTERS
      if (c.tid() == 0 && w.band_wave) ++emul_counters()[1];
#endif
      // no row with a bound (grid shortest path: every reduced row is a +a/-a pair): the one-wave elimination builds
      // the band rows it needs from the cone itself, the band is never written to memory (cone_band.h, BandGen)
      if (w.band_wave) {
        uint32_t nb = 0;
        double hd = 0.0;
        for (int i = c.tid(); i < p; i += C::NT) {
          nb += v.vkind[i] ? 0u : 1u;
          double s2 = 0.0;
          for (uint32_t e = v.mptr[i]; e < v.mptr[i + 1]; ++e) {
            const double val = v.pm1 ? 1.0 : (double)v.mval[e];
            s2 += val * val;
          }
          hd = fmax(hd, s2);
        }
        nb = c.reduce_add_u32(nb);
        hd = c.reduce_max(hd);
#ifdef CAVE_NO_BAND_GEN  // (diagnostic builds: keep the materialised band)
        nb = 1u;
#endif
        if (nb == 0u) {
          BandGen& gn = w.gen;
          gn.on = true;
          gn.mptr = v.mptr; gn.mcol = v.mcol; gn.mval = v.pm1 ? nullptr : v.mval;
          gn.cptr = v.cptr; gn.cvar = v.cvar; gn.cvalc = v.pm1 ? nullptr : v.cvalc;
          gn.usign = v.usign; gn.r = nullptr; gn.mu = 0.0; gn.hdiag = hd;
          rb_wanted = true;
#ifdef CAVE_EMUL_COUNTERS
          if (c.tid() == 0) ++emul_counters()[3];
#endif
        }
      }
#endif
      } else w.H = nullptr;
#if defined(CAVE_GPU_CODE) && !defined(CAVE_NO_RB)  // (diagnostic builds can pin the full-size band)
      if (rb_wanted && mode != MODE_IPM) {
        rb_setup(c, v, w, rb_cache, rb_cache_bytes);
#ifdef CAVE_EMUL_COUNTERS
        if (c.tid() == 0 && w.rb.on) ++emul_counters()[5];
#endif
      }
#endif
    } else {
      w.dn.on = false;
      w.gen.on = false;
      w.rb.on = false;
      if (p > C::PMAX) return ST_TOO_LARGE;
      w.hscale = 1.0;
      w.hinv = 1.0;
      if constexpr (C::NWAVES > 1) {  // fixed-point scale of the multi-wave Hessian sums: |H_ab| <= (largest entry)^2 * (longest row)
        double vm = v.pm1 ? 1.0 : 0.0, ml = 1.0;
        if (!v.pm1) {
          const uint32_t nz = v.mptr[p];
          for (uint32_t e = c.tid(); e < nz; e += (uint32_t)C::NT) vm = fmax(vm, fabs((double)v.mval[e]));
        }
        for (int i = c.tid(); i < p; i += C::NT) ml = fmax(ml, (double)(v.mptr[i + 1] - v.mptr[i]));
        vm = c.reduce_max(vm);
        ml = c.reduce_max(ml);
        w.hscale = fixed_scale(vm, vm * ml);
        w.hinv = 1.0 / w.hscale;
      }
      // rc[d], theta[32] and dv[32] are the always-zero dummies of the lite index structures (cone_core.h)
      const uint32_t pg = pp < 33u ? 33u : pp;
      w.rc = ar.get<double>((uint32_t)d + 1u);
      w.wold = ar.get<float>(d);
      w.theta = ar.get<double>(pg);
      w.ttry = ar.get<double>(pp);
      w.told = ar.get<double>(pp);
      w.g = ar.get<double>(pp);
      w.dv = ar.get<double>(pg);
      w.g2 = ar.get<double>(pp);
      w.step = ar.get<double>(pp);
      w.ldh = p | 1;
      w.tri = false;
      if constexpr (ctx_diet<C>::value) w.tri = v.csc_far;  // diet layout: H as the packed lower triangle
      w.H = ar.get<double>((uint32_t)(p > 0 ? (w.tri ? p * (p + 1) / 2 : p * w.ldh) : 1));
      w.act = ar.get<uint8_t>(pp);
    }
    w.res = res;
    w.q = tvec;  // the epilogue's target scratch is free while the solver runs
    uint8_t* lflag = ar.get<uint8_t>(pp);
    uint32_t* llist = ar.get<uint32_t>(pp);
    if (ar.ovf) return ST_TOO_LARGE;
    if constexpr (!LARGE) {
      if (c.tid() == 0) {
        w.rc[d] = 0.0;
        if (p <= 32) { w.theta[32] = 0.0; w.dv[32] = 0.0; }
      }
    }
    for (int i = c.tid(); i < p; i += C::NT) lflag[i] = (uint8_t)((v.mptr[i + 1] - v.mptr[i]) > kLongRow ? 1 : 0);
    c.sync();
    SolveView vv = v;
    vv.nlong = (int)c.compact_nonzero_u8(lflag, p, llist);
    vv.longrow = llist;
    c.sync();
    if constexpr (LARGE) {
      // dense reduced systems with g in LDS: the gradient is summed column-wise in fixed point (cone_dense.h dense_gradient)
      if (w.dn.on && hot && hot->owns(w.g) && C::WL > 1 && mode != MODE_IPM) vv.gcol_bound = gcol_bound;
    }
    if constexpr (LARGE) {  // (the streamed gradient shares very long rows between the waves)
      uint32_t* tlist = ar.get<uint32_t>(pp);
      if (ar.ovf) return ST_TOO_LARGE;
      for (int i = c.tid(); i < p; i += C::NT) lflag[i] = (uint8_t)((v.mptr[i + 1] - v.mptr[i]) > kTeamRow ? 1 : 0);
      c.sync();
      vv.nteam = (int)c.compact_nonzero_u8(lflag, p, tlist);
      vv.teamrow = tlist;
      c.sync();
    }
    SolveResult r;
    bool lite = false;
    if constexpr (LARGE) {
      if (mode == MODE_IPM) {  // interior-point inner mode on the band / dense forms (round 3)
        w.gen.on = false;  // the band is materialised: its weights are the smoothed clip's, not band_weight's
        for (int i = c.tid(); i < p; i += C::NT) w.act[i] = 0;
        c.sync();
        solve_cone_ipm_band_call(c, vv, w, max_iter, &r);
        lite = true;  // (handled)
      }
    } else if (mode == MODE_IPM) {
      r = solve_cone_ipm(c, vv, w, max_iter);
      lite = true;  // (handled)
    }
#if defined(CAVE_GPU_CODE)
    if constexpr (!LARGE && C::LITE_OK) if (!lite) {
      // small +-1 cone: Newton iteration on ONE wave over the lite index structures (cone_core.h).  Carried by
      // the kernel shapes with a 256-register budget (one and two waves per instance: it wants ~200 VGPRs, and
      // inlined into the 4-wave kernels, whose residency rests on 128, it cost 650 bytes of scratch per lane).
      // In the two-wave shape wave 1 helps with the scan and the build, then parks at the barrier below.
      // (A latency design: beyond ~2 instances per SIMD the general path's lower instruction count wins --
      //  packed TSP-20 at B = 4096: 374 us with the lite solver.)
      // Eligible: rows ordered [free | 0 .. 8 bound rows] (TSP: degree equalities, then cuts; grid cones: no bound
      // rows) and the epilogue's target scratch (idle while the solver runs) holds the reduced system -- the lite
      // solver eliminates the free rows once per Newton iteration and runs its active-set loop on the Schur
      // complement of the bound rows (lite_model_step).
      bool ls_ok = false;
      {
        uint32_t nfree = 0, bad = 0;
        for (int i = c.tid(); i < p; i += C::NT) nfree += v.vkind[i] ? 1u : 0u;
        nfree = c.reduce_add_u32(nfree);
        for (int i = c.tid(); i < p; i += C::NT) bad += ((v.vkind[i] != 0) != (i < (int)nfree)) ? 1u : 0u;
        bad = c.reduce_add_u32(bad);
        const int nI = p - (int)nfree;
        const uint32_t need = (uint32_t)(p * nI + nI * (nI | 1) + 4 * nI + (nI + 7) / 8);
        ls_ok = bad == 0u && nI <= 8 && need <= (uint32_t)d;
        w.ls_on = ls_ok;
        w.ls_nF = (int)nfree;
        w.ls_nI = nI;
        w.ls_scr = w.q;
      }
      LiteCone L;
      lite = gridDim.x <= 2048u && ls_ok && lite_build(c, ar, vv, L);
      if (lite) {
#ifdef CAVE_EMUL_COUNTERS
        if (c.tid() == 0) { ++emul_counters()[2]; if (w.ls_nI > 0) ++emul_counters()[4]; }
#endif
        r.f = 0.0; r.iters = 0; r.status = ST_OK;
        if (c.wave_id() == 0) {
          SoloCtx<32, 4> sc;
          sc.lane = c.lane_id();
          sc.lite = L;
#ifdef CAVE_STAMPS
          sc.st = c.st;
#endif
          r = solve_cone_impl<SoloCtx<32, 4>, true, false>(sc, vv, w, max_iter, 1e-11);
#ifdef CAVE_STAMPS
          c.st[13] += 1000000;  // diagnostic marker: the lite path ran
#endif
        }
        c.broadcast_from_wave0(r.f, r.iters, r.status);
      }
    }
#endif
    if constexpr (LARGE) CAVE_ACC(0);
    if (!lite) {
      if constexpr (LARGE) solve_cone_band_call(c, vv, w, max_iter, 1e-11, &r);
      else r = solve_cone<C, false>(c, vv, w, max_iter, 1e-11);
    }
    if constexpr (LARGE) CAVE_ACC(1);
    st = r.status;
    f = r.f;
    *iters_out = r.iters;
#if defined(CAVE_GPU_CODE)
    if constexpr (LARGE) {
      if (w.band_wave) {  // a wave gave up waiting for the other one: its rows were not the delivered ones
        c.sync();
        if (reinterpret_cast<const int*>(w.bwin + band_wave_flags_at(w.bw))[4] != 0) {
          if (warm_state && c.tid() == 0) *warm_state = 0;
          fill_failure(c, d, b, o);  // (nothing computed from undelivered rows leaves the kernel)
          return ST_NOT_CONVERGED;
        }
      }
    }
#endif
    if (warm_theta && warm_state) {  // keep the multipliers for the next solve of this cone
      if (st == ST_OK) for (int i = c.tid(); i < p; i += C::NT) warm_theta[i] = (float)w.theta[i];
      if (c.tid() == 0) *warm_state = (uint8_t)(st == ST_OK ? 1 : 0);
    }
    if (st == ST_BAD_INPUT) return st;
  } else if constexpr (LARGE) {
    res = ar.get<double>(d);
    tvec = ar.get<double>(d);
    if (ar.ovf) return ST_TOO_LARGE;
  }
  EpilogueOut eo;
  eo.proj = o.proj ? o.proj + b * d : nullptr;
  eo.rnorm = o.rnorm ? o.rnorm + b : nullptr;
  eo.target = o.target ? o.target + b * d : nullptr;
  eo.loss = o.loss ? o.loss + b : nullptr;
  eo.grad = o.grad ? o.grad + b * d : nullptr;
  epilogue(c, mode, d, sign, inner_ratio, empty, y, res, f, avg, tvec, eo);
  if constexpr (LARGE) CAVE_ACC(9);
  return st;
}

// LARGE: `ws` / `ws_bytes` is this workgroup's slice of the global workspace (the arena), LDS is the hot arena.
template <class C, bool LARGE = false>
CAVE_HD void run_dense_instance(C& c, unsigned char* smem, const DenseParams& P, int64_t b,
                                unsigned char* ws = nullptr, uint32_t ws_bytes = 0) {
  const int d = P.d, m = P.m;
  Arena ar, hot;
  hot.init(smem + C::SCRATCH_BYTES, P.lds_bytes - C::SCRATCH_BYTES);
  if constexpr (LARGE) ar.init(ws, ws_bytes);
  else ar = hot;
  ConeBuild cb;
  CAVE_T0();
  int32_t st = scan_and_build<C, LARGE>(c, ar, cb, P.ctrs + b * (int64_t)m * d, m, d, P.nnz_cap);
  CAVE_ACC(0);
  int iters = 0;
  if (st == ST_OK) {
    const bool need_avg = (P.mode == MODE_INNER || P.mode == MODE_HEURISTIC || P.mode == MODE_AVG);
    float* y = ar.get<float>(d);
    float* avg = need_avg ? ar.get<float>(d) : nullptr;
    if (ar.ovf) st = ST_TOO_LARGE;
    else {
      for (int k = c.tid(); k < d; k += C::NT) y[k] = P.pred ? P.sign * P.pred[b * d + k] : 0.f;
      c.sync();
      if (need_avg) compute_avg(c, cb, avg);
      ar.release_top();  // build-phase temporaries (row tags, unit counts, pair scratch) are dead now
      SolveView v = view_of(cb);
      CAVE_ACC(1);
      st = solve_and_finish<C, LARGE>(c, ar, LARGE ? &hot : nullptr, v, P.mode, P.sign, P.inner_ratio, P.max_iter, y,
                                      avg, b, P.o, &iters);
      CAVE_ACC(9);
    }
  }
  if (st == ST_TOO_LARGE || st == ST_BAD_INPUT) fill_failure(c, d, b, P.o);
  if (c.tid() == 0) {
    if (P.o.status) P.o.status[b] = st;
    if (P.o.iters) P.o.iters[b] = iters;
  }
}

template <class C, bool LARGE = false>
CAVE_HD void run_pack_instance(C& c, unsigned char* smem, const PackParams& P, int64_t b,
                               unsigned char* ws = nullptr, uint32_t ws_bytes = 0) {
  const int d = P.d, m = P.m;
  const int NT = C::NT;
  Arena ar;
  if constexpr (LARGE) ar.init(ws, ws_bytes);
  else ar.init(smem + C::SCRATCH_BYTES, P.lds_bytes - C::SCRATCH_BYTES);
  ConeBuild cb;
  CAVE_T0();
  int32_t st = scan_and_build<C, LARGE>(c, ar, cb, P.ctrs + b * (int64_t)m * d, m, d, P.nnz_cap);
  CAVE_ACC(0);
  if (!P.fill) {
    if (c.tid() == 0) {
      P.n_rows[b] = (st == ST_OK) ? cb.p : 0;
      P.n_nnz[b] = (st == ST_OK) ? (int32_t)cb.nnzM : 0;
      if (P.status) P.status[b] = st;
    }
    return;
  }
  const cave_cone_store& S = P.store;
  const int64_t slot = P.slot0 + b;
  float* avg = (st == ST_OK) ? ar.get<float>(d) : nullptr;
  if (st == ST_OK && ar.ovf) st = ST_TOO_LARGE;
  if (st == ST_OK) {
    const int64_t r0 = S.row_off[slot], z0 = S.nnz_off[slot];
    const bool slots = S.n_rows != nullptr;  // slot mode: the offsets are capacity windows
    // exact-fit store: it was sized from pass 1, refuse to write past it; slot mode: the instance must fit its window
    if (!slots && ((int64_t)cb.p != S.row_off[slot + 1] - r0 || (int64_t)cb.nnzM != S.nnz_off[slot + 1] - z0)) st = ST_BAD_INPUT;
    else if (slots && ((int64_t)cb.p > S.row_off[slot + 1] - r0 || (int64_t)cb.nnzM > S.nnz_off[slot + 1] - z0)) st = ST_TOO_LARGE;
    else {
      compute_avg(c, cb, avg);
      for (int k = c.tid(); k < d; k += NT) {
        S.usign[slot * d + k] = cb.usign[k];
        S.avg[slot * d + k] = avg[k];
      }
      for (int k = c.tid(); k <= d; k += NT) S.cptr[slot * (d + 1) + k] = cb.cptr[k];
      // reduced rows: extents, kinds, CSR entries (values expanded from the sign bit in +-1 mode)
      for (int i = c.tid(); i < cb.p; i += NT) {
        S.rlo[r0 + i] = cb.mptr[i];
        S.rhi[r0 + i] = cb.mptr[i + 1];
        S.vkind[r0 + i] = cb.vkind[i];
      }
      for (uint32_t e = c.tid(); e < cb.nnzM; e += NT) {
        const uint32_t x = cb.mcol[e];
        S.ccol[z0 + e] = (uint16_t)(cb.pm1 ? (x & 0x7fffu) : x);
        S.cval[z0 + e] = cb.pm1 ? ((x & 0x8000u) ? -1.0f : 1.0f) : cb.mval[e];
      }
      for (uint32_t e = c.tid(); e < cb.nnzM; e += NT) {
        const uint32_t x = cb.cvar[e];
        S.cvar[z0 + e] = (uint16_t)(cb.pm1 ? (x & 0x7fffu) : x);
        S.cvalc[z0 + e] = cb.pm1 ? ((x & 0x8000u) ? -1.0f : 1.0f) : cb.cvalc[e];
      }
      if (c.tid() == 0) {
        S.n_valid[slot] = cb.n_valid_proj;
        S.flags[slot] = (uint8_t)(cb.pm1 ? 1 : 0);
      }
    }
  }
  CAVE_ACC(1);
  if (S.n_rows != nullptr && c.tid() == 0) {
    S.n_rows[slot] = st == ST_OK ? cb.p : -1;  // -1: this slot holds no cone (the packed operator reports TOO_LARGE)
    S.n_nnz[slot] = st == ST_OK ? (int32_t)cb.nnzM : 0;
  }
  if (c.tid() == 0 && P.status) P.status[b] = st;
}

template <class C>
CAVE_HD void run_packed_instance(C& c, unsigned char* smem, const PackedParams& P, int64_t b) {
  const cave_cone_store& S = P.store;
  const int d = S.d;
  const int NT = C::NT;
  Arena ar;
  ar.init(smem + C::SCRATCH_BYTES, P.lds_bytes - C::SCRATCH_BYTES);
  const int64_t slot = P.ids ? P.ids[b] : b;
  int32_t st = ST_OK;
  int iters = 0;
  if (slot < 0 || slot >= S.n) st = ST_BAD_INPUT;
  else if (S.n_rows && S.n_rows[slot] < 0) st = ST_TOO_LARGE;  // slot mode: the cone did not fit its slot when packed
  else {
    const int64_t r0 = S.row_off[slot], z0 = S.nnz_off[slot];
    const int p = S.n_rows ? (int)S.n_rows[slot] : (int)(S.row_off[slot + 1] - r0);
    const uint32_t nz = S.n_nnz ? (uint32_t)S.n_nnz[slot] : (uint32_t)(S.nnz_off[slot + 1] - z0);
    const bool need_avg = (P.mode == MODE_INNER || P.mode == MODE_HEURISTIC || P.mode == MODE_AVG);
    const bool need_proj = (P.mode == MODE_PROJECT || P.mode == MODE_EXACT || P.mode == MODE_INNER || P.mode == MODE_IPM);
    const bool pm1 = (S.flags[slot] & 1) != 0;
    // Diet layout (contexts that carry it: the four-wave wide shape): taken by an instance whose ordinary arena does not
    // fit this launch's LDS -- the host then launched with the diet figure of its largest cone to get two workgroups per
    // compute unit (TSP-50: 107 KB -> 76 KB).  Needs the signs folded into the store's indices (flags bit 1).
    bool diet = false;
    if constexpr (ctx_diet<C>::value) {
      diet = pm1 && (S.flags[slot] & 2) != 0 && P.mode != MODE_IPM && p > kLiteMaxRows &&
             (uint32_t)packed_lds_bytes(d, p, (int32_t)nz, true, false, false) > P.lds_bytes;
    }
    float* y = ar.get<float>(d);
    float* avg = (need_avg && !diet) ? ar.get<float>(d) : nullptr;
    uint8_t* usign = ar.get<uint8_t>(d);
    uint32_t* cptr = ar.get<uint32_t>(d + 1);
    uint32_t* mptr = ar.get<uint32_t>((uint32_t)p + 1u);
    uint8_t* vkind = ar.get<uint8_t>(p > 0 ? p : 1);
    uint16_t* mcol = ar.get<uint16_t>(nz > 0 ? nz : 1);
    uint16_t* cvar = diet ? nullptr : ar.get<uint16_t>(nz > 0 ? nz : 1);
    float* mval = pm1 ? nullptr : ar.get<float>(nz > 0 ? nz : 1);
    float* cvalc = pm1 ? nullptr : ar.get<float>(nz > 0 ? nz : 1);
    if (ar.ovf) st = ST_TOO_LARGE;
    else {
      // The instance comes from global memory (HBM / L2: ~1 k cycles per dependent load): every loop below requests
      // U rounds of operands before it stores any, so a 530-entry TSP-20 cone costs three memory round trips instead
      // of nine (measured: the load was 15 k of the one-wave kernel's 32 k cycles outside the Newton loop).
      constexpr int U = 4;
      const bool stage_avg = need_avg && !diet;
      for (int k0 = c.tid(); k0 < d; k0 += U * NT) {
        float yv[U], av[U];
        uint8_t uv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = k0 + u * NT, kc = k < d ? k : d - 1;
          yv[u] = P.pred ? P.pred[b * d + kc] : 0.f;
          uv[u] = S.usign[slot * d + kc];
          av[u] = stage_avg ? S.avg[slot * d + kc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = k0 + u * NT;
          if (k < d) {
            y[k] = P.sign * yv[u];
            usign[k] = uv[u];
            if (stage_avg) avg[k] = av[u];
          }
        }
      }
      if (need_proj) {
        for (int k0 = c.tid(); k0 <= d; k0 += U * NT) {
          uint32_t cv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) { const int k = k0 + u * NT; cv[u] = S.cptr[slot * (d + 1) + (k <= d ? k : d)]; }
#pragma unroll
          for (int u = 0; u < U; ++u) { const int k = k0 + u * NT; if (k <= d) cptr[k] = cv[u]; }
        }
        for (int i = c.tid(); i < p; i += NT) {
          mptr[i] = S.rlo[r0 + i];
          vkind[i] = S.vkind[r0 + i];
        }
        if (c.tid() == 0) mptr[p] = nz;
        const uint32_t elast = nz > 0u ? nz - 1u : 0u;
        for (uint32_t e0 = c.tid(); e0 < nz; e0 += (uint32_t)(U * NT)) {
          uint16_t cc[U], cr[U];
          float cvl[U], crl[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t e = e0 + (uint32_t)(u * NT), ec = e < nz ? e : elast;  // clamped: loads are unconditional
            cc[u] = S.ccol[z0 + ec];
            cvl[u] = S.cval[z0 + ec];
            cr[u] = diet ? (uint16_t)0 : S.cvar[z0 + ec];
            crl[u] = diet ? 0.f : S.cvalc[z0 + ec];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t e = e0 + (uint32_t)(u * NT);
            if (e < nz) {
              if (pm1) {
                // (stores of the large path / the diet tier carry the sign in bit 15 already: flags bit 1)
                mcol[e] = (uint16_t)((cc[u] & 0x7fffu) | (cvl[u] < 0.f ? 0x8000u : 0u));
                if (!diet) cvar[e] = (uint16_t)((cr[u] & 0x7fffu) | (crl[u] < 0.f ? 0x8000u : 0u));
              } else {
                mcol[e] = cc[u];
                mval[e] = cvl[u];
                cvar[e] = cr[u];
                cvalc[e] = crl[u];
              }
            }
          }
        }
      }
      c.sync();
      SolveView v;
      v.d = d; v.p = p; v.n_valid = S.n_valid[slot]; v.pm1 = pm1;
      v.mptr = mptr; v.mcol = mcol; v.mval = mval; v.vkind = vkind;
      v.cptr = cptr; v.cvar = diet ? S.cvar + z0 : cvar; v.cvalc = cvalc; v.usign = usign;
      v.nlong = 0; v.longrow = nullptr;
      v.csc_far = diet;
      st = solve_and_finish(c, ar, nullptr, v, P.mode, P.sign, P.inner_ratio, P.max_iter, y,
                            (need_avg && diet) ? S.avg + slot * d : avg, b, P.o, &iters,
                            S.warm_theta ? S.warm_theta + r0 : nullptr, S.warm_state ? S.warm_state + slot : nullptr);
    }
  }
  if (st == ST_TOO_LARGE || st == ST_BAD_INPUT) fill_failure(c, d, b, P.o);
  if (c.tid() == 0) {
    if (P.o.status) P.o.status[b] = st;
    if (P.o.iters) P.o.iters[b] = iters;
  }
}

// Large-cone form: the SolveView points straight into the store (global memory); only the row
// pointers (the store keeps per-row extents) and the work arrays are carved from the workspace / LDS.
template <class C>
CAVE_HD void run_packed_large_instance(C& c, unsigned char* smem, const PackedParams& P, int64_t b, unsigned char* ws,
                                       uint32_t ws_bytes) {
  const cave_cone_store& S = P.store;
  const int d = S.d;
  const int NT = C::NT;
  Arena ar, hot;
  ar.init(ws, ws_bytes);
  hot.init(smem + C::SCRATCH_BYTES, P.lds_bytes - C::SCRATCH_BYTES);
  const int64_t slot = P.ids ? P.ids[b] : b;
  int32_t st = ST_OK;
  int iters = 0;
  if (slot < 0 || slot >= S.n) st = ST_BAD_INPUT;
  else if (S.n_rows && S.n_rows[slot] < 0) st = ST_TOO_LARGE;  // slot mode: the cone did not fit its slot when packed
  else {
    const int64_t r0 = S.row_off[slot], z0 = S.nnz_off[slot];
    const int p = S.n_rows ? (int)S.n_rows[slot] : (int)(S.row_off[slot + 1] - r0);
    const uint32_t nz = S.n_nnz ? (uint32_t)S.n_nnz[slot] : (uint32_t)(S.nnz_off[slot + 1] - z0);
    const bool need_avg = (P.mode == MODE_INNER || P.mode == MODE_HEURISTIC || P.mode == MODE_AVG);
    float* y = ar.get<float>(d);
    uint32_t* mptr = ar.get<uint32_t>((uint32_t)p + 1u);
    if (ar.ovf) st = ST_TOO_LARGE;
    else {
      for (int k = c.tid(); k < d; k += NT) y[k] = P.pred ? P.sign * P.pred[b * d + k] : 0.f;
      for (int i = c.tid(); i < p; i += NT) mptr[i] = S.rlo[r0 + i];
      if (c.tid() == 0) mptr[p] = nz;
      c.sync();
      SolveView v;
      // flags bit 1 (set by the host for stores of this path): the indices of this all-+-1 instance carry the signs in
      // bit 15, the value arrays are not read
      v.d = d; v.p = p; v.n_valid = S.n_valid[slot]; v.pm1 = (S.flags[slot] & 2) != 0;
      v.mptr = mptr; v.mcol = S.ccol + z0; v.mval = S.cval + z0; v.vkind = S.vkind + r0;
      v.cptr = S.cptr + slot * (d + 1); v.cvar = S.cvar + z0; v.cvalc = S.cvalc + z0; v.usign = S.usign + slot * d;
      v.nlong = 0; v.longrow = nullptr;
      const float* avg = need_avg ? S.avg + slot * d : nullptr;
      st = solve_and_finish<C, true>(c, ar, &hot, v, P.mode, P.sign, P.inner_ratio, P.max_iter, y, avg, b, P.o, &iters,
                                     S.warm_theta ? S.warm_theta + r0 : nullptr,
                                     S.warm_state ? S.warm_state + slot : nullptr,
                                     S.rb_cache ? S.rb_cache + slot * S.rb_stride : nullptr,
                                     S.rb_cache ? (uint64_t)S.rb_stride : 0ull);
    }
  }
  if (st == ST_TOO_LARGE || st == ST_BAD_INPUT) fill_failure(c, d, b, P.o);
  if (c.tid() == 0) {
    if (P.o.status) P.o.status[b] = st;
    if (P.o.iters) P.o.iters[b] = iters;
  }
}

// ---- arena sizing shared by host code of both builds
static inline uint32_t align8u(uint64_t x) { return (uint32_t)((x + 7u) & ~7ull); }
static constexpr uint32_t kMaxLds = 160u * 1024u;

// Upper bound of the arena a dense launch needs, assuming at most `rows_raw` general rows,
// `p` reduced rows and `nnzM` reduced non-zeros (the kernel reports ST_TOO_LARGE otherwise).
// `nt` = threads per workgroup (per-thread dump slots of the scan), `pm1` = +-1 cones (no value arrays).
// `lite_room`: reserve the index structures of the one-wave lite solver (one- and two-wave launches of up to 2048
// instances use it; without the room lite_build declines and the general solver runs).
static inline uint64_t arena_bytes_dense(int64_t m, int64_t d, int64_t cap, int64_t rows_raw, int64_t p, int64_t nnzM,
                                         int64_t nt = 256, bool pm1 = false, bool lite_room = true) {
  // bottom: persistent through the solve
  uint64_t persist = align8u(d) + align8u(4 * (d + 1)) + align8u(4 * (p + 1)) + align8u(p)       // usign, cptr, mptr, vkind
                     + 2 * align8u(2 * nnzM) + (pm1 ? 0 : 2 * align8u(4 * nnzM));               // mcol, cvar (+ mval, cvalc)
  // top: scan output + build-phase temporaries
  uint64_t scan = 2 * align8u(4 * (cap + nt)) + align8u(4 * (m + 1));                            // erc, eall, rptr
  uint64_t temps = align8u(4 * d) + align8u(4 * m) + align8u(m) + align8u(4 * (cap / 64 + 1))    // ucnt, rs2, rowtag, long rows
                   + 3 * align8u(4 * rows_raw) + align8u(rows_raw) + align8u(2 * m)               // vraw, vnorm, twin, keep, rowvar
                   + (2 * align8u(8 * rows_raw) > align8u(4 * d) ? 2 * align8u(8 * rows_raw) : align8u(4 * d));  // signatures | fill
  uint64_t vecs = 2 * align8u(4 * d);                                                            // y, avg
  uint64_t solve = 3 * align8u(8 * d) + align8u(4 * d)                                           // res, tvec/q, rc, wold
                   + 7 * align8u(8 * p) + align8u(8 * p * (p | 1)) + 2 * align8u(p) + align8u(4 * p)  // theta..step, told, H, act, long rows
                   + 8 + 2 * 8 * 33;                                                              // dummy slots of the lite form
  if (lite_room && pm1 && p <= kLiteMaxRows && d <= kLiteMaxD) solve += lite_lds_bytes((int)d, (uint32_t)nnzM);
  uint64_t build_peak = persist + scan + temps + vecs;
  uint64_t solve_peak = persist + vecs + solve;
  return (build_peak > solve_peak ? build_peak : solve_peak) + 64 + 256;  // + context scratch
}

static inline int32_t default_limits(int64_t m_max, int64_t d, int32_t* nnz_cap, int32_t* lds_bytes, bool lite_room = true) {
  // structured cones: <= d unit entries + a few sparse rows; dense tiny cones: m*d
  int64_t cap = 4 * (m_max + d) + 256;
  if (cap > m_max * d) cap = m_max * d;
  if (cap < 64) cap = 64;
  // sized for structured cones (+-1 entries, <= 32 reduced rows, <= 64 general rows); a cone that
  // needs more reports ST_TOO_LARGE and the caller retries with the full 160 KiB arena
  uint64_t need = arena_bytes_dense(m_max, d, cap, 64, 32, cap * 6 / 10, 256, true, lite_room);
  while (need > kMaxLds && cap > 256) {
    cap = cap * 3 / 4;
    need = arena_bytes_dense(m_max, d, cap, 64, 32, cap * 6 / 10, 256, true, lite_room);
  }
  if (need > kMaxLds) need = kMaxLds;
  if (nnz_cap) *nnz_cap = (int32_t)cap;
  if (lds_bytes) *lds_bytes = (int32_t)need;
  return 0;
}

static inline bool resolve_limits(int64_t m, int64_t d, int32_t& cap, int32_t& lds, bool lite_room = true) {
  int32_t dcap = 0, dlds = 0;
  default_limits(m, d, &dcap, &dlds, lite_room);
  if (cap <= 0) cap = dcap;
  if (lds <= 0) {
    // honour a caller-supplied nnz_cap when deriving the arena size
    uint64_t need = arena_bytes_dense(m, d, cap, 64, 32, (int64_t)cap * 6 / 10, 256, true, lite_room);
    lds = (int32_t)(need > kMaxLds ? kMaxLds : need);
  }
  return lds > 0 && (uint32_t)lds <= kMaxLds && cap > 0;
}

// ---- large-cone path: bytes of global workspace one workgroup needs (its arena).
// `band` = p * (bw + 1) entries of the band Hessian (kept twice: H and its factor).
static inline uint64_t large_slice_bytes(int64_t m, int64_t d, int64_t cap, int64_t band) {
  const int64_t rows_raw = m, p = m < 65534 ? m : 65534, nnzM = cap;
  uint64_t persist = align8u(d) + align8u(4 * (d + 1)) + align8u(4 * (p + 1)) + align8u(p) + 2 * align8u(2 * nnzM) +
                     2 * align8u(4 * nnzM);
  uint64_t scan = 2 * (uint64_t)align8u(4 * cap) + align8u(4 * (m + 1));
  uint64_t temps = align8u(4 * d) + align8u(4 * m) + align8u(m) + align8u(4 * (cap / 64 + 1)) + 3 * align8u(4 * rows_raw) +
                   align8u(rows_raw) + align8u(2 * m) +
                   (2 * align8u(8 * rows_raw) > align8u(4 * d) ? 2 * align8u(8 * rows_raw) : align8u(4 * d)) + align8u(4 * p);
  uint64_t vecs = 2 * align8u(4 * d);
  uint64_t solve = 3 * align8u(8 * d) + align8u(d) + 8 * align8u(8 * p) + 2 * align8u(p) + 2 * align8u(4 * p) +
                   3 * (uint64_t)8 * (uint64_t)(band + 1) + 2 * 8 * 4096;  // H, factor, ring window ((bw+1)^2 <= p*(bw+1)), staging
  uint64_t build_peak = persist + scan + temps + vecs;
  uint64_t solve_peak = persist + vecs + solve;
  return (build_peak > solve_peak ? build_peak : solve_peak) + 256;
}
static inline uint64_t packed_large_slice_bytes(int64_t d, int64_t max_rows, int64_t band) {
  const int64_t p = max_rows > 0 ? max_rows : 1;
  return align8u(4 * d) + align8u(4 * (p + 1)) + 3 * align8u(8 * d) + align8u(d) + 8 * align8u(8 * p) + 2 * align8u(p) +
         2 * align8u(4 * p) + 3 * (uint64_t)8 * (uint64_t)(band + 1) + 2 * 8 * 4096 + 256;  // + staging buffers (at most 4096 entries each)
}

// LDS for the hot arrays of the large-cone path (solve_and_finish<LARGE> allocates in this order: window, z, step,
// act, staging; what is left takes further row vectors).  Narrow bands: exactly what the one-wave elimination
// needs, so that four workgroups fit a CU; otherwise a 4 KiB-rounded figure with room for a third row vector.
static inline uint32_t packed_large_lds_bytes(int64_t max_rows, int64_t max_bw) {
  const uint64_t p = max_rows > 0 ? max_rows : 1, ld = (uint64_t)max_bw + 1u;
  if (dense_shape((int)(p > 0x7fffffff ? 0x7fffffff : p), (int)max_bw)) {
    // dense reduced systems (cone_dense.h): the folded matrix, its scratch, the Schur system of up to 32 bound rows,
    // and the row vectors of the Newton iteration -- 72 KB at p = 105, so two workgroups share a CU
    const uint64_t nI = p < (uint64_t)kDenseMaxBound ? p : (uint64_t)kDenseMaxBound;
    uint64_t tot = dense_lds_bytes((int)p, (int)nI) + 64u + 9u * (8u * p + 8u) + p + 256u + 64u;
    tot = (tot + 255u) & ~255ull;
    if (tot <= kMaxLds) return (uint32_t)tot;
  }
  if (band_wave_fits((int)max_bw, (int)(p > 0x7fffffff ? 0x7fffffff : p))) {
    const uint64_t need = 8ull * (band_wave_region((int)max_bw, (int)(p > 0x7fffffff ? 0x7fffffff : p)) + 2ull * p) + p + 64u;
    const uint64_t tot = ((need + 256u + 64u) + 255u) & ~255ull;  // + context scratch, alignment slack
    if (tot * 4u <= kMaxLds) return (uint32_t)tot;
  }
  uint64_t want = 8ull * (ld * (ld + 2u) + 2u * 32u * ld + 3u * p) + p + 4096u;
  want = (want + 4095u) / 4096u * 4096u;
  if (want < 32u * 1024u) want = 32u * 1024u;
  return (uint32_t)(want > kMaxLds ? kMaxLds : want);
}

// lite_room: reserve the index structures of the one-wave lite solver (only launches of up to 2048 instances use it)
// diet: the layout for +-1 cones of the TSP-50 class (50 - 64 reduced rows, thousands of non-zeros: 100+ KB otherwise,
// one workgroup per compute unit) -- H as a packed lower triangle, the CSC entries and the average normal read in
// place from the packed store instead of staged: ~30 KB less, two workgroups per compute unit.
CAVE_HOSTDEV int32_t packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz, bool all_pm1, bool lite_room, bool diet) {
  uint64_t s = 0;
  int64_t p = max_rows;
  const uint64_t a8 = 7u;
  auto al = [&](uint64_t x) -> uint64_t { return (x + a8) & ~a8; };
  s += al(4 * d) * (diet ? 1 : 2) + al(d) + al(4 * (d + 1));                              // y, (avg,) usign, cptr
  s += al(4 * (p + 1)) + al(p);                                                           // mptr, vkind
  s += (diet ? 1 : 2) * (al(2 * (int64_t)max_nnz) + (all_pm1 ? 0 : al(4 * (int64_t)max_nnz)));  // CSR (+ CSC) (+ values)
  s += al(8 * d) * 3 + al(4 * d);                                                         // res, tvec, rc, wold
  s += 7 * al(8 * p) + (diet ? al(8 * (p * (p + 1) / 2)) : al(8 * p * (p | 1))) + 2 * al(p) + al(4 * p) + 128 + 256 + 8 + 2 * 8 * 33;
  if (lite_room && !diet && all_pm1 && p <= kLiteMaxRows && d <= kLiteMaxD) {  // lite index structures, if they fit (optional)
    const uint64_t with_lite = s + lite_lds_bytes((int)d, (uint32_t)max_nnz);
    if (with_lite <= 160u * 1024u) s = with_lite;
  }
  if (s > 160u * 1024u) return -1;
  return (int32_t)s;
}
static inline int32_t packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz, bool all_pm1 = false,
                                       bool lite_room = true) {
  return packed_lds_bytes(d, max_rows, max_nnz, all_pm1, lite_room, false);
}

}  // namespace cave
