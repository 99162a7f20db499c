// cone_step.h — the two halves of the fused "step" kernel (kernels.h cone_step_kernel) for small +-1 cones on the
// dense wire format (TSP-20, small grids: what the one-wave lite solver of cone_core.h takes).
//
// A training step on the reference's dense format is  scan + cone build (depends on the cones only)  ->  Newton solve
// + loss / gradient (needs the prediction).  The reference hands the cones of batch i+1 to the loop before the
// predictor has produced the prediction of batch i+1 (the DataLoader collates ahead, src/dataset.py:133-144), so the
// first half of batch i+1 can run beside the second half of batch i.  Here both halves sit in ONE grid:
//   blocks [0, B)            solve instance b of the CURRENT batch from a transient "lite store" (one wave each)
//   blocks [B, B + B_next)   pack instance b - B of the NEXT dense batch into the other lite store (four waves each)
// Blocks are dispatched in index order, so the solve waves take their SIMDs first and the pack workgroups fill what is
// left of each compute unit: no second stream, no event, no dependence on how the runtime maps streams to hardware
// queues (the side-stream form of rounds 2-3 needed a spacer kernel to win a dispatch race).
//
// The lite store holds, per instance, exactly what the one-wave solver reads, in the layout it reads it (the `ell` and
// `csr16` index structures of cone_core.h are built ONCE, by the pack half, instead of by every solve).
#pragma once
#include "../../include/cave_hip.h"
#include "cone_common.h"
#include "cone_core.h"
#include "cone_instance.h"

namespace cave {

static constexpr int kLiteHdr = 8;           // int32 words per slot: see cave_lite_store::hdr
static constexpr int kLiteCsrWords = 32 * kLiteMaxChunk;  // csr16 stride (uint32 words) per slot
static constexpr uint32_t kStepElectBytes = 64;           // head of the LDS block: words of the wave election

struct StepSolveParams {
  cave_lite_store store;
  const int64_t* ids;   // store slot of batch entry b (null: slot b -- the transient per-batch stores)
  const float* pred;
  int64_t B;
  int32_t mode;
  float sign, inner_ratio;
  int32_t max_iter;
  int32_t flags;        // CAVE_STEP_ZERO_FAILED: a failed instance gets loss 0 and a zero gradient instead of NaN / its last iterate
  OutPtrs o;
};

struct StepPackParams {
  const float* ctrs;
  int64_t B;
  int32_t m, d;
  uint32_t nnz_cap;
  cave_lite_store store;
  int32_t* status;
};

struct StepParams {
  StepSolveParams S;
  StepPackParams Q;
  uint32_t lds_bytes;   // dynamic LDS of the launch (both halves size their arenas from it)
  uint32_t* tickets;    // [4096] per-compute-unit SIMD claim masks of the wave election (caller-owned, zeroed once)
};

// LDS one solve block needs for cost dimension d (reduced systems of up to 32 rows, up to 1536 non-zeros)
static inline uint32_t step_solve_lds_bytes(int64_t d) {
  const uint64_t p = kLiteMaxRows;
  uint64_t s = kStepElectBytes;
  s += 2 * align8u(4 * d) + align8u(d) + align8u(4 * (p + 1)) + align8u(p);   // y, avg, usign, mptr, vkind
  s += 16 + 16 * (uint64_t)d + 16 + 4 * (uint64_t)kLiteCsrWords + 8 * (33 + 64 + 65) + 40;  // ell, csr16, rs, rl
  s += 2 * align8u(8 * d) + align8u(8 * (d + 1)) + align8u(4 * d);             // res, tvec, rc, wold
  s += 2 * 8 * 33 + 5 * align8u(8 * p) + align8u(8 * p * (p | 1)) + align8u(p) + 64;  // theta, dv, 5 vectors, H, act
  return (uint32_t)((s + 255u) & ~255ull);
}

// -------------------------------------------------------------------------------------------------- pack half
// Eligibility + lite index structures of one cone (SolveView in LDS, `avg` its average normal) -> slot `slot` of the
// lite store.  hdr[0] = 1: the slot holds a cone the one-wave solver takes;  -1: it does not (not +-1, more than 32
// reduced rows / 8 entries per column / 8 bound rows, rows not ordered [free | bound], or no room in the arena): the
// solve half reports CAVE_ST_TOO_LARGE for it and the host falls back to the general operator.  Returns the state.
template <class C>
CAVE_HD int32_t write_lite_slot(C& c, Arena& ar, const SolveView& v, const float* avg, uint32_t nnzM, const cave_lite_store& S,
                                int64_t slot) {
  const int NT = C::NT;
  const int d = v.d, p = v.p;
  int nF = 0;
  bool ok = v.pm1 && p <= kLiteMaxRows && d <= kLiteMaxD;
  LiteCone L;
  L.ell = nullptr; L.csr16 = nullptr; L.rs = nullptr; L.rl = nullptr; L.chn8 = 0; L.cmax = 0;
  if (ok && p > 0) {
    uint32_t nfree = 0, bad = 0;
    for (int i = c.tid(); i < p; i += NT) nfree += v.vkind[i] ? 1u : 0u;
    nfree = c.reduce_add_u32(nfree);
    for (int i = c.tid(); i < p; i += NT) bad += ((v.vkind[i] != 0) != (i < (int)nfree)) ? 1u : 0u;
    bad = c.reduce_add_u32(bad);
    const int nI = p - (int)nfree;
    ok = bad == 0u && nI <= 8;  // (the solve half finds room for lite_model_step's scratch or reports TOO_LARGE)
    nF = (int)nfree;
    if (ok) ok = lite_build(c, ar, v, L);
  }
  if (ok) {
    for (int k = c.tid(); k < d; k += NT) {
      S.usign[slot * d + k] = v.usign[k];
      S.avg[slot * d + k] = avg[k];
    }
    for (int i = c.tid(); i <= p; i += NT) S.rowptr[slot * (kLiteMaxRows + 1) + i] = v.mptr[i];
    if (p > 0) {
      // 16-byte copies: both structures are 16-byte aligned in LDS and in the store
      const uint4* e4 = reinterpret_cast<const uint4*>(L.ell);
      uint4* eo = reinterpret_cast<uint4*>(S.ell + slot * 4 * (int64_t)d);
      for (int k = c.tid(); k < d; k += NT) eo[k] = e4[k];
      const uint4* c4 = reinterpret_cast<const uint4*>(L.csr16);
      uint4* co = reinterpret_cast<uint4*>(S.csr16 + slot * (int64_t)kLiteCsrWords);
      for (int k = c.tid(); k < 8 * L.chn8; k += NT) co[k] = c4[k];
      for (int i = c.tid(); i < p; i += NT) S.rl[slot * kLiteMaxRows + i] = L.rl[i];
    }
  }
  if (c.tid() == 0) {
    int32_t* h = S.hdr + slot * kLiteHdr;
    h[0] = ok ? 1 : -1;
    h[1] = ok ? p : 0;
    h[2] = ok ? (int32_t)nnzM : 0;
    h[3] = nF;
    h[4] = ok ? v.n_valid : 0;
    h[5] = L.cmax;
    h[6] = L.chn8;
    h[7] = 0;
  }
  return ok ? 1 : -1;
}

// scan + cone build (as run_pack_instance), then the lite slot
template <class C>
CAVE_HD void run_pack_lite_instance(C& c, unsigned char* smem, uint32_t lds_bytes, const StepPackParams& P, int64_t b) {
  const int d = P.d, m = P.m;
  Arena ar;
  ar.init(smem + C::SCRATCH_BYTES, lds_bytes - C::SCRATCH_BYTES);
  ConeBuild cb;
  CAVE_T0();
  int32_t st = scan_and_build<C, false, true>(c, ar, cb, P.ctrs + b * (int64_t)m * d, m, d, P.nnz_cap);
  CAVE_ACC(0);
  const cave_lite_store& S = P.store;
  float* avg = (st == ST_OK) ? ar.get<float>(d) : nullptr;
  if (st == ST_OK && ar.ovf) st = ST_TOO_LARGE;
  if (st == ST_OK) {
    compute_avg(c, cb, avg);
    ar.release_top();  // build-phase temporaries are dead now
    const SolveView v = view_of(cb);
    if (write_lite_slot(c, ar, v, avg, cb.nnzM, S, b) != 1) st = ST_TOO_LARGE;
  } else if (c.tid() == 0) {
    int32_t* h = S.hdr + b * kLiteHdr;
    h[0] = -1;
    for (int i = 1; i < kLiteHdr; ++i) h[i] = 0;
  }
  CAVE_ACC(1);
  if (c.tid() == 0 && P.status) P.status[b] = st;
}

// One instance of a packed cone store (cave_cone_store) -> its lite slot: cones are static per instance
// (src/dataset.py:72), so a device-resident store builds the solver's index structures ONCE, when it is created.
struct LiteFromPackedParams {
  cave_cone_store src;
  cave_lite_store dst;
  int64_t n;
  uint32_t lds_bytes;
  int32_t* status;
};
template <class C>
CAVE_HD void run_lite_from_packed(C& c, unsigned char* smem, const LiteFromPackedParams& P, int64_t slot) {
  const cave_cone_store& S = P.src;
  const int d = S.d, NT = C::NT;
  Arena ar;
  ar.init(smem + C::SCRATCH_BYTES, P.lds_bytes - C::SCRATCH_BYTES);
  const int64_t r0 = S.row_off[slot], z0 = S.nnz_off[slot];
  const int p = S.n_rows ? (int)S.n_rows[slot] : (int)(S.row_off[slot + 1] - r0);
  const uint32_t nz = S.n_nnz ? (uint32_t)S.n_nnz[slot] : (uint32_t)(S.nnz_off[slot + 1] - z0);
  const bool pm1 = (S.flags[slot] & 1) != 0;
  int32_t state = -1;
  if (p >= 0 && p <= kLiteMaxRows && pm1 && d <= kLiteMaxD && nz <= 64u * (uint32_t)kLiteMaxChunk) {
    float* avg = ar.get<float>(d);
    uint8_t* usign = ar.get<uint8_t>(d);
    uint32_t* cptr = ar.get<uint32_t>(d + 1);
    uint32_t* mptr = ar.get<uint32_t>((uint32_t)p + 1u);
    uint8_t* vkind = ar.get<uint8_t>(p > 0 ? p : 1);
    uint16_t* mcol = ar.get<uint16_t>(nz > 0 ? nz : 1);
    uint16_t* cvar = ar.get<uint16_t>(nz > 0 ? nz : 1);
    if (!ar.ovf) {
      for (int k = c.tid(); k < d; k += NT) { avg[k] = S.avg[slot * d + k]; usign[k] = S.usign[slot * d + k]; }
      for (int k = c.tid(); k <= d; k += NT) cptr[k] = S.cptr[slot * (d + 1) + k];
      for (int i = c.tid(); i < p; i += NT) { mptr[i] = S.rlo[r0 + i]; vkind[i] = S.vkind[r0 + i]; }
      if (c.tid() == 0) mptr[p] = nz;
      for (uint32_t e = c.tid(); e < nz; e += NT) {  // sign into bit 15, as the LDS-resident solvers keep +-1 cones
        mcol[e] = (uint16_t)((S.ccol[z0 + e] & 0x7fffu) | (S.cval[z0 + e] < 0.f ? 0x8000u : 0u));
        cvar[e] = (uint16_t)((S.cvar[z0 + e] & 0x7fffu) | (S.cvalc[z0 + e] < 0.f ? 0x8000u : 0u));
      }
      c.sync();
      SolveView v;
      v.d = d; v.p = p; v.n_valid = S.n_valid[slot]; v.pm1 = true;
      v.mptr = mptr; v.mcol = mcol; v.mval = nullptr; v.vkind = vkind;
      v.cptr = cptr; v.cvar = cvar; v.cvalc = nullptr; v.usign = usign;
      v.nlong = 0; v.longrow = nullptr;
      state = write_lite_slot(c, ar, v, avg, nz, P.dst, slot);
    }
  }
  if (state != 1 && c.tid() == 0) {
    int32_t* h = P.dst.hdr + slot * kLiteHdr;
    h[0] = -1;
    for (int i = 1; i < kLiteHdr; ++i) h[i] = 0;
  }
  if (c.tid() == 0 && P.status) P.status[slot] = state == 1 ? ST_OK : ST_TOO_LARGE;
}

// ------------------------------------------------------------------------------------------------- solve half
#if defined(CAVE_GPU_CODE)
// One wave: load slot b of the lite store into LDS (every load of the prologue is issued before the first store:
// one memory round trip), run the one-wave Newton solver, fused epilogue.  `lane`: 0..63.
template <class SC>
CAVE_HD void run_lite_instance(SC& sc, unsigned char* smem, uint32_t lds_bytes, const StepSolveParams& P, int64_t b) {
  const cave_lite_store& S = P.store;
  const int d = S.d;
  const int lane = sc.lane;
  Arena ar;
  ar.init(smem, lds_bytes);
  int32_t st = ST_OK;
  int iters = 0;
  const int64_t slot_raw = P.ids ? P.ids[b] : b;
  const bool in_range = slot_raw >= 0 && slot_raw < S.n;
  const int64_t slot = in_range ? slot_raw : 0;
  const int mode = P.mode;
  const bool need_avg = (mode == MODE_INNER || mode == MODE_HEURISTIC || mode == MODE_AVG);
  const bool need_proj = (mode == MODE_PROJECT || mode == MODE_EXACT || mode == MODE_INNER);
  // ---- prologue: EVERY global load of the instance in one memory round trip -- the header words beside the arrays
  // (nothing below depends on them: a slot's arrays have fixed extents, what a cone does not use is loaded and dropped);
  // until round 4 the header came first and the arrays a latency later (~2 us of a 70 us instance)
  constexpr int KC = (kLiteMaxD + 63) / 64;  // coordinates per lane
  float yv[KC], av[KC];
  uint4 ev[KC];
  uint8_t uv[KC];
#pragma unroll
  for (int s = 0; s < KC; ++s) {
    const int k = lane + 64 * s, kc = k < d ? k : d - 1;
    yv[s] = P.pred ? P.pred[b * d + kc] : 0.f;
    uv[s] = S.usign[slot * d + kc];
    av[s] = need_avg ? S.avg[slot * d + kc] : 0.f;
    ev[s] = need_proj ? reinterpret_cast<const uint4*>(S.ell + slot * 4 * (int64_t)d)[kc] : make_uint4(0, 0, 0, 0);
  }
  uint4 cv[kLiteMaxChunk / 8];
#pragma unroll
  for (int g8 = 0; g8 < kLiteMaxChunk / 8; ++g8)
    cv[g8] = need_proj ? reinterpret_cast<const uint4*>(S.csr16 + slot * (int64_t)kLiteCsrWords)[g8 * 64 + lane] : make_uint4(0, 0, 0, 0);
  const uint32_t mp_raw = (need_proj && lane <= kLiteMaxRows) ? S.rowptr[slot * (kLiteMaxRows + 1) + lane] : 0u;
  const uint8_t rl_raw = (need_proj && lane < kLiteMaxRows) ? S.rl[slot * kLiteMaxRows + lane] : (uint8_t)0;
  const int32_t* hdr = S.hdr + slot * kLiteHdr;
  const int32_t hv = lane < kLiteHdr ? hdr[lane] : 0;  // (one load; the words are handed out below)
  const int32_t state = in_range ? __builtin_amdgcn_readlane(hv, 0) : 0;
  const int p = __builtin_amdgcn_readlane(hv, 1), nF = __builtin_amdgcn_readlane(hv, 3), n_valid = __builtin_amdgcn_readlane(hv, 4);
  const int cmax = __builtin_amdgcn_readlane(hv, 5), chn8 = __builtin_amdgcn_readlane(hv, 6);
  if (!in_range) st = ST_BAD_INPUT;
  else if (state != 1 || p < 0 || p > kLiteMaxRows || chn8 > kLiteMaxChunk || mode == MODE_IPM) st = ST_TOO_LARGE;
  else {
    const uint32_t pp = (uint32_t)(p > 0 ? p : 1);
    float* y = ar.get<float>(d);
    float* avg = need_avg ? ar.get<float>(d) : nullptr;
    uint8_t* usign = ar.get<uint8_t>(d);
    uint32_t* mptr = ar.get<uint32_t>(pp + 1u);
    uint8_t* vkind = ar.get<uint8_t>(pp);
    uint32_t* ell = ar.try_get<uint32_t, 16u>(4u * (uint32_t)d);
    uint32_t* csr16 = ar.try_get<uint32_t, 16u>(32u * (uint32_t)(chn8 > 0 ? chn8 : 8));
    double* rs = ar.get<double>(33u + 64u + 65u);
    uint8_t* rl = ar.get<uint8_t>(40u);
    double* res = ar.get<double>(d);
    double* tvec = ar.get<double>(d);
    SolveWork w;
    w.y = y;
    w.res = res;
    w.q = tvec;
    w.rc = ar.get<double>((uint32_t)d + 1u);
    w.wold = ar.get<float>(d);
    w.theta = ar.get<double>(33u);
    w.dv = ar.get<double>(33u);
    w.ttry = ar.get<double>(pp);
    w.told = ar.get<double>(pp);
    w.g = ar.get<double>(pp);
    w.g2 = ar.get<double>(pp);
    w.step = ar.get<double>(pp);
    w.ldh = p | 1;
    w.H = ar.get<double>((uint32_t)(p > 0 ? p * w.ldh : 1));
    w.act = ar.get<uint8_t>(pp);
    if (ar.ovf || !ell || !csr16) st = ST_TOO_LARGE;
    else {
      // ---- the LDS stores of what the prologue loaded
      const uint32_t mp = lane <= p ? mp_raw : 0u;
      const uint8_t rlv = lane < p ? rl_raw : (uint8_t)0;
#pragma unroll
      for (int s = 0; s < KC; ++s) {
        const int k = lane + 64 * s;
        if (k < d) {
          y[k] = P.sign * yv[s];
          usign[k] = uv[s];
          if (need_avg) avg[k] = av[s];
          if (need_proj && p > 0) reinterpret_cast<uint4*>(ell)[k] = ev[s];
        }
      }
#pragma unroll
      for (int g8 = 0; g8 < kLiteMaxChunk / 8; ++g8)
        if (need_proj && g8 * 8 < chn8) reinterpret_cast<uint4*>(csr16)[g8 * 64 + lane] = cv[g8];
      if (lane <= p) mptr[lane] = mp;
      if (lane < p) { rl[lane] = rlv; vkind[lane] = (uint8_t)(lane < nF ? 1 : 0); }
      if (lane == 0) {
        rs[33 + 64 + 64] = 0.0;
        w.rc[d] = 0.0;
        w.theta[32] = 0.0;
        w.dv[32] = 0.0;
      }
      sc.sync();
      SolveView v;
      v.d = d; v.p = p; v.n_valid = n_valid; v.pm1 = true;
      v.mptr = mptr; v.mcol = nullptr; v.mval = nullptr; v.vkind = vkind;
      v.cptr = nullptr; v.cvar = nullptr; v.cvalc = nullptr; v.usign = usign;
      v.nlong = 0; v.longrow = nullptr;
      const bool empty = (n_valid == 0);
      double f = 0.0;
      // scratch of lite_model_step: the epilogue's target vector (idle while the solver runs) when it is big enough,
      // else a block of its own (small cost dimensions: the arena is sized for d = 256)
      const int nI = p - nF;
      const uint32_t need = (uint32_t)(p * nI + nI * (nI | 1) + 4 * nI + (nI + 7) / 8);
      double* scr = need <= (uint32_t)d ? w.q : ar.try_get<double>(need);
      if (need_proj && !empty && !scr) st = ST_TOO_LARGE;
      else if (need_proj && !empty) {
        w.ls_on = true;
        w.ls_nF = nF;
        w.ls_nI = nI;
        w.ls_scr = scr;
        w.warm = nullptr;
        w.bw = 0; w.band_wave = false; w.band_hot = false; w.bwin = nullptr; w.bfac = nullptr; w.bz = nullptr; w.bstg = nullptr; w.bch = 0;
        w.dn.on = false;
        w.gen.on = false;
        sc.lite.ell = ell; sc.lite.csr16 = csr16; sc.lite.rs = rs; sc.lite.rl = rl; sc.lite.chn8 = chn8; sc.lite.cmax = cmax;
        const SolveResult r = solve_cone_impl<SC, true, false>(sc, v, w, P.max_iter, 1e-11);
        st = r.status;
        f = r.f;
        iters = r.iters;
      }
      if (st != ST_BAD_INPUT && st != ST_TOO_LARGE) {
        EpilogueOut eo;
        eo.proj = P.o.proj ? P.o.proj + b * d : nullptr;
        eo.rnorm = P.o.rnorm ? P.o.rnorm + b : nullptr;
        eo.target = P.o.target ? P.o.target + b * d : nullptr;
        eo.loss = P.o.loss ? P.o.loss + b : nullptr;
        eo.grad = P.o.grad ? P.o.grad + b * d : nullptr;
        epilogue(sc, mode, d, P.sign, P.inner_ratio, empty, y, res, f, avg, tvec, eo);
      }
    }
  }
  if (st == ST_TOO_LARGE || st == ST_BAD_INPUT) fill_failure(sc, d, b, P.o);
  if (st != ST_OK && (P.flags & CAVE_STEP_ZERO_FAILED)) {
    // training with the status examined later (check='lazy'): the failed instance must not reach the optimizer
    if (P.o.grad) for (int k = lane; k < d; k += 64) P.o.grad[b * d + k] = 0.f;
    if (P.o.loss && lane == 0) P.o.loss[b] = 0.f;
  }
  if (lane == 0) {
    if (P.o.status) P.o.status[b] = st;
    if (P.o.iters) P.o.iters[b] = iters;
  }
}
#endif  // CAVE_GPU_CODE

}  // namespace cave
