// cone_rb.h — red-black reduction of the Newton systems of band cones without bound rows (grid shortest path).
//
// The reduced rows of such a cone are the nodes of a graph whose edges are the coordinates (a coordinate's column holds
// the two rows it joins); H = M W M^T is its weighted Laplacian-like matrix, and the band LDL^T pays a dependent chain
// per row: 900 pivots for a 30 x 30 grid on one eliminator wave (cone_band.h), ~75 % of an instance's time.  An
// INDEPENDENT set of rows R -- no two of them share a coordinate, so H_RR is diagonal: every other node of a grid --
// needs no elimination at all:
//     x_R = D^-1 (g_R - H_RB x_B),      (H_BB - H_BR D^-1 H_RB) x_B = g_B - H_BR D^-1 g_R,      D = diag(H_RR).
// The Schur complement S on the other rows B keeps the half bandwidth in the order of B (two rows of a grid apart = the
// same distance in B's numbering), so the band solver runs unchanged on HALF the rows: half the pivots of the
// factorisation and of both substitutions, half the factor traffic.
//
// Taken by +-1 cones whose coordinates join at most two rows and whose rows hold at most four such coordinates.  Per call
// (the cone is static, the weights are not): the adjacency (row -> four (coordinate, neighbour, sign) slots); R = the
// greedy independent set by row index (row i joins when no earlier row of R is its neighbour: a checkerboard on a grid
// in row-major order), found by parallel rounds on states in LDS; and a RECIPE for every row of S: its entries as
// (coordinate, coordinate, red row, offset, signs) records, so that an iteration costs two memory levels per row -- the
// records, then the weights they name -- instead of a seven-level walk through the cone.  Per Newton iteration: the
// weights, the diagonal, the rows of S (one thread per row in a fixed order: deterministic, no atomics; staged through
// the idle LDS ring, written to the workspace block such cones leave unused), the band solve on the black rows, the red
// rows in closed form.
#pragma once
#include "cone_common.h"

namespace cave {

constexpr int kRbDeg = 4;  // neighbour slots per row
constexpr uint32_t kRbRecPerRow = 12;  // records the persistent cache holds per reduced row (a grid needs ~5)

// Bytes of the per-instance cache (cave_cone_store.rb_cache) for cones of up to `p` reduced rows: what the reduction
// derives from the STATIC cone alone -- header (state, black rows, records), black rows, recipe pointers, the red and
// the black neighbour slots, positions, classes, the records -- so that only an instance's first projection builds it.
CAVE_HOSTDEV uint64_t rb_cache_bytes(int64_t p) {
  if (p < 8 || p > 0x7fff) return 0;
  const uint64_t P = (uint64_t)p;
  auto a16 = [](uint64_t x) { return (x + 15ull) & ~15ull; };
  return 64ull + a16(4 * P) + a16(4 * (P + 1)) + 2 * a16(4ull * kRbDeg * P) + a16(2 * P) + a16(P) + a16(8ull * kRbRecPerRow * P);
}

#if defined(CAVE_GPU_CODE)
// adjacency slot: coordinate (bit 15: sign of the product of the two entries of its column) | neighbour << 16;
// 0xffffffff = empty
// `cache` / `cache_bytes`: this instance's block of the store's persistent cache (rb_cache_bytes), or null: header word 0
// = 0 nothing yet, 1 built (word 1: black rows), 2 this cone does not take the reduction.
template <class C>
CAVE_HD void rb_setup(C& c, const SolveView& v, SolveWork& w, unsigned char* cache = nullptr, uint64_t cache_bytes = 0) {
  constexpr int NT = C::NT;
  const int p = v.p, d = v.d, ldh = w.ldh, bw = w.bw, tid = c.tid();
  RbWork& rb = w.rb;
  rb.on = false;
  if (cache && cache_bytes < rb_cache_bytes(p)) cache = nullptr;
  // ---- carve: per-call arrays in the workspace block of the (never materialised) band, the persistent ones in the cache
  // when there is one (else behind them)
  unsigned char* base = reinterpret_cast<unsigned char*>(w.H);
  const uint64_t room = 8ull * (uint64_t)p * (uint64_t)ldh;
  uint64_t off = 0;
  auto take = [&](uint64_t bytes) { unsigned char* q = base + off; off += (bytes + 15ull) & ~15ull; return q; };
  rb.wt = reinterpret_cast<double*>(take(8ull * d));
  rb.hd = reinterpret_cast<double*>(take(8ull * p));
  rb.hdB = reinterpret_cast<double*>(take(8ull * p));
  rb.gB = reinterpret_cast<double*>(take(8ull * p));
  uint32_t* adjG_ = reinterpret_cast<uint32_t*>(take(4ull * kRbDeg * p));  // adjacency slots (set-up only)
  unsigned char* pbase = base;
  uint64_t poff = off, proom = room;
  if (cache) { pbase = cache; poff = 64; proom = cache_bytes; }
  auto ptake = [&](uint64_t bytes) { unsigned char* q = pbase + poff; poff += (bytes + 15ull) & ~15ull; return q; };
  rb.blk = reinterpret_cast<uint32_t*>(ptake(4ull * p));
  rb.rp = reinterpret_cast<uint32_t*>(ptake(4ull * (p + 1)));
  rb.radj = reinterpret_cast<uint32_t*>(ptake(4ull * kRbDeg * p));
  rb.badj = reinterpret_cast<uint32_t*>(ptake(4ull * kRbDeg * p));
  rb.pos = reinterpret_cast<uint16_t*>(ptake(2ull * p));
  rb.cls = reinterpret_cast<uint8_t*>(ptake(1ull * p));
  rb.rec = reinterpret_cast<uint32_t*>(ptake(0));
  if (off >= room || poff >= proom) return;
  volatile uint32_t* hdr = reinterpret_cast<volatile uint32_t*>(cache);
  if (cache) {
    // one thread reads the state, the workgroup agrees on it (another workgroup solving the same instance may be
    // writing it right now: whoever sees "built" finds complete data behind the fence, the others build the same)
    uint32_t stt = 0, nb = 0;
    if (tid == 0) { stt = hdr[0]; nb = hdr[1]; }
    stt = c.reduce_add_u32(stt);
    nb = c.reduce_add_u32(nb);
    if (stt == 2u) return;
    if (stt == 1u) {
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL)
      __threadfence();
#endif
      rb.nB = (int)nb;
      rb.on = true;
      return;
    }
  }
  auto give_up = [&]() {  // remember that this cone does not take the reduction
    if (cache && tid == 0) hdr[0] = 2u;
  };
  if (!v.pm1 || p < 8 || p > 0x7fff || d > 0x7fff) return give_up();
  if (p > 32 * NT) return give_up();  // (a thread keeps its undecided rows in one word below)
  const auto mptr = space_cast<1>(v.mptr);
  const auto mcol = space_cast<1>(v.mcol);
  const auto cptr = space_cast<1>(v.cptr);
  const auto cvar = space_cast<1>(v.cvar);
  // a coordinate joins at most two rows, a row holds at most kRbDeg coordinates
  uint32_t bad = 0;
  struct LdE { uint32_t lo, hi; };
  strided_batched<8, NT>(tid, d, [&](int k) { return LdE{cptr[k], cptr[k + 1]}; },
                         [&](int, const LdE& x) { bad += (x.hi - x.lo > 2u) ? 1u : 0u; });
  strided_batched<8, NT>(tid, p, [&](int i) { return LdE{mptr[i], mptr[i + 1]}; },
                         [&](int, const LdE& x) { bad += (x.hi - x.lo > (uint32_t)kRbDeg) ? 1u : 0u; });
  if (c.reduce_add_u32(bad) != 0u) return give_up();
  // LDS (the idle ring of the band solver) holds what the set-up walks again and again: states [p] bytes, positions and
  // black rows [p] halfwords each, the neighbours [p][kRbDeg] halfwords, row pointers of the recipes [p + 1] words (first
  // the compacted list of black rows).  17 p bytes; the workspace receives the results and the full adjacency slots.
  const uint64_t lds_room = 8ull * band_wave_flags_at(bw);
  const uint32_t pos_at = ((uint32_t)p + 15u) & ~15u;
  const uint32_t blk_at = (pos_at + 2u * (uint32_t)p + 15u) & ~15u;
  const uint32_t oth_at = (blk_at + 2u * (uint32_t)p + 15u) & ~15u;
  const uint32_t rp_at = (oth_at + 2u * kRbDeg * (uint32_t)p + 15u) & ~15u;
  if ((uint64_t)rp_at + 4ull * (p + 1) > lds_room) return;
  unsigned char* lbase = reinterpret_cast<unsigned char*>(w.bwin);
  auto st = space_cast<3>(reinterpret_cast<uint8_t*>(lbase));
  auto lpos = space_cast<3>(reinterpret_cast<uint16_t*>(lbase + pos_at));
  auto lblk = space_cast<3>(reinterpret_cast<uint16_t*>(lbase + blk_at));
  auto oth = space_cast<3>(reinterpret_cast<uint16_t*>(lbase + oth_at));  // 0xffff: no neighbour in this slot
  auto lrp = space_cast<3>(reinterpret_cast<uint32_t*>(lbase + rp_at));
  const auto adjG = space_cast<1>(adjG_);
  // ---- adjacency: four rows per thread in flight (extents, entries, column extents, column entries)
  // slot: coordinate (bit 15: sign of the product of the two entries of its column) | neighbour << 16; 0xffffffff = empty
  constexpr int RA = 4;
  for (int i0 = tid; i0 < p; i0 += RA * NT) {
    uint32_t lo[RA], n[RA], kx[RA][kRbDeg], clo[RA][kRbDeg], ccnt[RA][kRbDeg], x0[RA][kRbDeg], x1[RA][kRbDeg];
    const uint32_t elast = mptr[p] > 0u ? mptr[p] - 1u : 0u;
#pragma unroll
    for (int u = 0; u < RA; ++u) {
      const int i = i0 + u * NT;
      const int ic = i < p ? i : p - 1;
      lo[u] = mptr[ic];
      n[u] = i < p ? mptr[ic + 1] - lo[u] : 0u;
    }
#pragma unroll
    for (int u = 0; u < RA; ++u)
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s) kx[u][s] = mcol[lo[u] + (uint32_t)s < elast ? lo[u] + (uint32_t)s : elast];
#pragma unroll
    for (int u = 0; u < RA; ++u)
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s) {
        clo[u][s] = cptr[kx[u][s] & 0x7fffu];
        ccnt[u][s] = cptr[(kx[u][s] & 0x7fffu) + 1u] - clo[u][s];
      }
    const uint32_t clast = cptr[d] > 0u ? cptr[d] - 1u : 0u;
#pragma unroll
    for (int u = 0; u < RA; ++u)
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s) {
        x0[u][s] = cvar[clo[u][s] < clast ? clo[u][s] : clast];
        x1[u][s] = cvar[clo[u][s] + 1u < clast ? clo[u][s] + 1u : clast];
      }
#pragma unroll
    for (int u = 0; u < RA; ++u) {
      const int i = i0 + u * NT;
      if (i >= p) continue;
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s) {
        uint32_t slot = 0xffffffffu;
        if ((uint32_t)s < n[u] && ccnt[u][s] == 2u) {
          const bool first = (x0[u][s] & 0x7fffu) == (uint32_t)i;
          const uint32_t other = (first ? x1[u][s] : x0[u][s]) & 0x7fffu;
          const uint32_t sg = ((x0[u][s] ^ x1[u][s]) & 0x8000u);  // sign of the product of the column's two entries
          slot = (kx[u][s] & 0x7fffu) | sg | (other << 16);
        }
        adjG[i * kRbDeg + s] = slot;
        oth[i * kRbDeg + s] = (uint16_t)(slot >> 16);
      }
      st[i] = 0;
    }
  }
  c.sync();
  // ---- greedy independent set by row index, parallel rounds: a row turns black as soon as an earlier neighbour is red,
  // red once all its earlier neighbours are black (a state only ever goes from 0 to its final value: a stale read
  // delays a decision by a round, it never changes it).  A thread walks only the rows it has not decided yet (bit u of
  // `open` = row tid + u * NT): on a grid a row is decided in the round its anti-diagonal comes up, so a round costs a
  // couple of LDS reads and a barrier; whether everybody is done is asked every eighth round.
  uint32_t open = 0;
  for (int u = 0; u < 32; ++u) open |= (tid + u * NT < p) ? (1u << u) : 0u;
  constexpr int RPT = 8;  // rows per thread whose earlier neighbours are cached in registers (p <= RPT * NT)
  if (p <= RPT * NT) {
    // the earlier neighbours of this thread's rows, read once (0xffff: none): a round is then one batch of state reads
    // (59 rounds on a 30 x 30 grid at ~3 k cycles each: ~90 us of a 3.5 ms instance -- with the checkerboard written
    // directly, as a timing experiment, the kernel took 3.43 ms instead of 3.51; the rounds are instruction-bound)
    uint16_t low[RPT][kRbDeg];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int i = tid + u * NT;
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s) {
        const uint32_t o = i < p ? (uint32_t)oth[i * kRbDeg + s] : 0xffffu;
        low[u][s] = (uint16_t)((o < (uint32_t)i) ? o : 0xffffu);
      }
    }
    for (int round = 0; round <= p + 8; ++round) {
      uint8_t x[RPT][kRbDeg];
      uint32_t live = 0;  // (wave-uniform) the row groups some lane of this wave still has open
#pragma unroll
      for (int u = 0; u < RPT; ++u) live |= (__ballot((open >> u) & 1u) != 0ull) ? (1u << u) : 0u;
#pragma unroll
      for (int u = 0; u < RPT; ++u)
        if ((live >> u) & 1u) {
#pragma unroll
          for (int s = 0; s < kRbDeg; ++s) x[u][s] = (((open >> u) & 1u) && low[u][s] != 0xffffu) ? st[low[u][s]] : (uint8_t)2;
        }
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        if (!((open >> u) & 1u)) continue;
        bool wait = false, red_nb = false;
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) {
          red_nb = red_nb || x[u][s] == 1;
          wait = wait || x[u][s] == 0;
        }
        if (red_nb) { st[tid + u * NT] = 2; open &= ~(1u << u); }
        else if (!wait) { st[tid + u * NT] = 1; open &= ~(1u << u); }
      }
      c.sync();
      if ((round & 7) == 7 && c.reduce_add_u32(open != 0u ? 1u : 0u) == 0u) break;
    }
  } else
  for (int round = 0; round <= p + 8; ++round) {
    uint32_t m = open;
    while (m) {
      const int u = __builtin_ctz(m);
      m &= m - 1u;
      const int i = tid + u * NT;
      bool wait = false, red_nb = false;
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s) {
        const uint32_t other = oth[i * kRbDeg + s];
        if (other >= (uint32_t)i) continue;  // (0xffff: empty)
        const uint8_t x = st[other];
        red_nb = red_nb || x == 1;
        wait = wait || x == 0;
      }
      if (red_nb) { st[i] = 2; open &= ~(1u << u); }
      else if (!wait) { st[i] = 1; open &= ~(1u << u); }
    }
    c.sync();
    if ((round & 7) == 7 && c.reduce_add_u32(open != 0u ? 1u : 0u) == 0u) break;
  }
  const auto cls = space_cast<1>(rb.cls);
  const auto pos = space_cast<1>(rb.pos);
  const auto blk = space_cast<1>(rb.blk);
  const int nB = (int)c.compact_mask_u8(reinterpret_cast<const uint8_t*>(lbase), p, 0xff, 2, reinterpret_cast<uint32_t*>(lbase + rp_at));
  c.sync();
  rb.nB = nB;
  if (nB >= p || nB < 1 || !band_wave_fits(bw, nB)) return give_up();
  for (int i = tid; i < p; i += NT) cls[i] = st[i];
  for (int q = tid; q < nB; q += NT) {
    const uint32_t b = lrp[q];
    lblk[q] = (uint16_t)b;
    blk[q] = b;
    pos[b] = (uint16_t)q;
    lpos[b] = (uint16_t)q;
  }
  c.sync();
  // ---- recipes.  Row q of S (black row b):  a black neighbour o through coordinate k adds s w_k at column pos(o);
  // a red neighbour r through k1, and r's neighbour o2 through k2, subtract (s1 w_k1 / D_r)(s2 w_k2) at column pos(o2)
  // (o2 = b itself: the diagonal).  Only the columns >= q are kept (upper band).  Counted from the LDS copies, filled from
  // the adjacency slots (the row's, then its red neighbours': two memory levels per row).
  const auto rp = space_cast<1>(rb.rp);
  const auto radj = space_cast<1>(rb.radj);
  for (int q = tid; q < nB; q += NT) {
    const uint32_t b = lblk[q];
    uint32_t cnt = 0;
#pragma unroll
    for (int s = 0; s < kRbDeg; ++s) {
      const uint32_t o1 = oth[b * kRbDeg + s];
      if (o1 == 0xffffu) continue;
      if (st[o1] == 2) { cnt += ((int)lpos[o1] > q) ? 1u : 0u; continue; }
#pragma unroll
      for (int s2 = 0; s2 < kRbDeg; ++s2) {
        const uint32_t o2 = oth[o1 * kRbDeg + s2];
        if (o2 != 0xffffu) cnt += ((int)lpos[o2] >= q) ? 1u : 0u;
      }
    }
    lrp[q] = cnt;
  }
  if (tid == 0) lrp[nB] = 0u;
  c.sync();
  const uint32_t nrec = c.exclusive_scan_u32(reinterpret_cast<uint32_t*>(lbase + rp_at), nB + 1);
  if (poff + 8ull * (uint64_t)nrec > proom) return give_up();
  for (int q = tid; q <= nB; q += NT) rp[q] = lrp[q];
  const auto rec = space_cast<1>(rb.rec);
  const auto badj = space_cast<1>(rb.badj);
  double span = 0.0;
  constexpr int RF = 2;  // rows per thread in flight: their slots, then their red neighbours' slots (two memory levels)
  for (int q0 = tid; q0 < nB; q0 += RF * NT) {
    uint32_t bb[RF], a1[RF][kRbDeg], a2[RF][kRbDeg][kRbDeg];
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      bb[u] = lblk[q0 + u * NT < nB ? q0 + u * NT : nB - 1];
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s) a1[u][s] = adjG[bb[u] * kRbDeg + s];
    }
#pragma unroll
    for (int u = 0; u < RF; ++u)
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s)
#pragma unroll
        for (int s2 = 0; s2 < kRbDeg; ++s2) a2[u][s][s2] = adjG[(a1[u][s] == 0xffffffffu ? bb[u] : (a1[u][s] >> 16)) * kRbDeg + s2];
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      const int q = q0 + u * NT;
      if (q >= nB) continue;
      uint32_t at = lrp[q];
      auto emit = [&](uint32_t k1, uint32_t k2, uint32_t r, uint32_t offs, uint32_t first) {
        rec[2u * at] = k1 | (k2 << 16);
        rec[2u * at + 1u] = r | ((offs & 0xffu) << 16) | (first << 24);
        span = fmax(span, (double)offs);
        ++at;
      };
#pragma unroll
      for (int s = 0; s < kRbDeg; ++s) {
        const uint32_t a = a1[u][s];
        const uint32_t o1 = a >> 16;
        uint32_t bslot = 0xffffffffu;  // (coordinate | sign, red neighbour) of the black row at position q
        if (a != 0xffffffffu) {
          if (st[o1] == 2) {
            const int j = (int)lpos[o1];
            if (j > q) emit(a & 0xffffu, 0xffffu, 0u, (uint32_t)(j - q), 0u);
          } else {
            bslot = a;
            uint32_t first = 1u;
#pragma unroll
            for (int s2 = 0; s2 < kRbDeg; ++s2) {
              const uint32_t b2 = a2[u][s][s2];
              if (b2 == 0xffffffffu) continue;
              const int j = (int)lpos[b2 >> 16];
              if (j >= q) { emit(a & 0xffffu, b2 & 0xffffu, o1, (uint32_t)(j - q), first); first = 0u; }
            }
          }
        }
        badj[q * kRbDeg + s] = bslot;
      }
    }
  }
  // red rows: (coordinate | sign, position of the black neighbour) per slot
  struct LdA { uint32_t a[kRbDeg]; };
  strided_batched<4, NT>(tid, p, [&](int i) { LdA x; for (int s = 0; s < kRbDeg; ++s) x.a[s] = adjG[i * kRbDeg + s]; return x; },
                         [&](int i, const LdA& x) {
#pragma unroll
                           for (int s = 0; s < kRbDeg; ++s) {
                             const uint32_t a = x.a[s];
                             radj[i * kRbDeg + s] = (a == 0xffffffffu || st[i] != 1) ? 0xffffffffu : ((a & 0xffffu) | ((uint32_t)lpos[a >> 16] << 16));
                           }
                         });
  const int bwS = (int)c.reduce_max(span);
  c.sync();
  if (bwS > bw || bwS > 255) return give_up();
  rb.on = true;
  if (cache) {  // publish: the data first, then the state
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL)
    __threadfence();
#endif
    c.sync();
    if (tid == 0) {
      hdr[1] = (uint32_t)nB;
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL)
      __threadfence();
#endif
      hdr[0] = 1u;
    }
  }
}

// One Newton system: (H + shift I) step = rhs, H = M W M^T with this iteration's weights (w.gen.r, w.gen.mu).
template <class C>
CAVE_HD void rb_solve(C& c, const SolveView& v, SolveWork& w, const double* rhs, double reg_rel) {
  constexpr int NT = C::NT;
  const int p = v.p, d = v.d, ldh = w.ldh, bw = w.bw, tid = c.tid();
  const RbWork& rb = w.rb;
  const int nB = rb.nB;
  const auto mptr = space_cast<1>(v.mptr);
  const auto mcol = space_cast<1>(v.mcol);
  const auto usign = space_cast<1>(v.usign);
  const auto cls = space_cast<1>(rb.cls);
  const auto pos = space_cast<1>(rb.pos);
  const auto wt = space_cast<1>(rb.wt);
  const auto hd = space_cast<1>(rb.hd);
  const auto hdB = space_cast<1>(rb.hdB);
  const auto badj = space_cast<1>(rb.badj);
  const auto gB = space_cast<1>(rb.gB);
  const auto radj = space_cast<1>(rb.radj);
  const auto blk = space_cast<1>(rb.blk);
  const double* r = w.gen.r;
  const double inv_mu = w.gen.mu;
  const double shift = reg_rel * w.gen.hdiag;
  {  // weights: eight coordinates per thread in flight
    constexpr int R = 8;
    for (int k0 = tid; k0 < d; k0 += R * NT) {
      uint8_t us[R];
      double rk[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int kc = k0 + u * NT < d ? k0 + u * NT : d - 1;
        us[u] = usign[kc];
        rk[u] = r[kc];
      }
#pragma unroll
      for (int u = 0; u < R; ++u)
        if (k0 + u * NT < d) wt[k0 + u * NT] = band_weight(us[u], rk[u], inv_mu);
    }
  }
  c.sync();
  // diagonal of H + shift (entries are +-1): eight rows per thread in flight
  {
    constexpr int R = 8;
    const uint32_t elast = mptr[p] > 0u ? mptr[p] - 1u : 0u;
    for (int i0 = tid; i0 < p; i0 += R * NT) {
      uint32_t lo[R], n[R], kx[R][kRbDeg], cl[R], ps[R];
      double wk[R][kRbDeg];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int i = i0 + u * NT;
        const int ic = i < p ? i : p - 1;
        lo[u] = mptr[ic];
        n[u] = i < p ? mptr[ic + 1] - lo[u] : 0u;
        cl[u] = cls[ic];
        ps[u] = pos[ic];
      }
#pragma unroll
      for (int u = 0; u < R; ++u)
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) kx[u][s] = mcol[lo[u] + (uint32_t)s < elast ? lo[u] + (uint32_t)s : elast] & 0x7fffu;
#pragma unroll
      for (int u = 0; u < R; ++u)
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) wk[u][s] = wt[kx[u][s]];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int i = i0 + u * NT;
        if (i >= p) continue;
        double s2 = shift;
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s)
          if ((uint32_t)s < n[u]) s2 += wk[u][s];
        if (cl[u] == 1) hd[i] = (s2 > 1e-300) ? 1.0 / s2 : 0.0;
        else { hd[i] = s2; hdB[ps[u]] = s2; }
      }
    }
  }
  c.sync();
  // reduced right-hand side: four rows per thread in flight
  {
    constexpr int R = 4;
    for (int q0 = tid; q0 < nB; q0 += R * NT) {
      uint32_t bb[R], ba[R][kRbDeg];
      double g0[R], wk[R][kRbDeg], dr[R][kRbDeg], gr[R][kRbDeg];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int qc = q0 + u * NT < nB ? q0 + u * NT : nB - 1;
        bb[u] = blk[qc];
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) ba[u][s] = badj[qc * kRbDeg + s];
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        g0[u] = rhs[bb[u]];
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) {
          const bool on = ba[u][s] != 0xffffffffu;
          const uint32_t k1 = on ? (ba[u][s] & 0x7fffu) : 0u, rr = on ? (ba[u][s] >> 16) : 0u;
          wk[u][s] = wt[k1];
          dr[u][s] = hd[rr];
          gr[u][s] = rhs[rr];
        }
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int q = q0 + u * NT;
        if (q >= nB) continue;
        double acc = g0[u];
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) {
          if (ba[u][s] == 0xffffffffu) continue;
          const double h = (ba[u][s] & 0x8000u) ? -wk[u][s] : wk[u][s];
          acc -= (h * dr[u][s]) * gr[u][s];
        }
        gB[q] = acc;
      }
    }
  }
  c.sync();
  // the band solver on the black rows: its producer builds the rows of S from the recipes (cone_band.h band_gen_rows_rb)
  BandGen gen = w.gen;
  gen.rb = &w.rb;
  solve_spd_band_wave<C::NWAVES>(c.lane_id(), c.wave_id(), nullptr, bw, rb.gB, w.act, nB, 0.0, w.bwin, w.bfac, w.bz, w.step, &gen,
#ifdef CAVE_STAMPS
                                 c.st,
#else
                                 nullptr,
#endif
                                 p);  // (x and the LDS region were sized for p rows)
  c.sync();
  // back to the order of the reduced rows: black rows copy, red rows follow in closed form
  for (int q = tid; q < nB; q += NT) w.bz[q] = w.step[q];
  c.sync();
  {
    constexpr int R = 4;
    for (int i0 = tid; i0 < p; i0 += R * NT) {
      uint32_t cl[R], ps[R], ra[R][kRbDeg];
      double hv[R], gv[R], wk[R][kRbDeg];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int ic = i0 + u * NT < p ? i0 + u * NT : p - 1;
        cl[u] = cls[ic];
        ps[u] = pos[ic];
        hv[u] = hd[ic];
        gv[u] = rhs[ic];
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) ra[u][s] = radj[ic * kRbDeg + s];
      }
#pragma unroll
      for (int u = 0; u < R; ++u)
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) wk[u][s] = wt[ra[u][s] == 0xffffffffu ? 0u : (ra[u][s] & 0x7fffu)];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int i = i0 + u * NT;
        if (i >= p) continue;
        if (cl[u] == 2) { w.step[i] = w.bz[ps[u]]; continue; }
        double acc = gv[u];
#pragma unroll
        for (int s = 0; s < kRbDeg; ++s) {
          if (ra[u][s] == 0xffffffffu) continue;
          const double h = (ra[u][s] & 0x8000u) ? -wk[u][s] : wk[u][s];
          acc -= h * w.bz[ra[u][s] >> 16];
        }
        w.step[i] = hv[u] * acc;
      }
    }
  }
  c.sync();
}
#endif

}  // namespace cave
