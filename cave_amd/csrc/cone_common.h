// cone_common.h — types shared by the HIP kernels and the serial test build.
//
// Everything in csrc/ is written once against a small "SPMD context" (Ctx)
// interface: the HIP build instantiates it with one 64-lane wavefront per
// training instance (ctx_wave.h); tests/emul instantiates it with a single
// serial lane so the identical control flow can be run under gcc + ASan/UBSan
// on a machine without a GPU.  The serial build is test infrastructure only —
// the Python package never loads it.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#if defined(__HIPCC__)
#define CAVE_HD __device__ __forceinline__
#define CAVE_NOINLINE __device__ __noinline__  // own register allocation (cone_band.h)
#define CAVE_HOSTDEV __host__ __device__ inline
#else
#define CAVE_HOSTDEV inline
#define CAVE_HD inline
#define CAVE_NOINLINE inline
#endif

// CAVE_GPU_CODE: the wave-level code (DPP / readlane / ballot: the one-wave lite solver, the one-/two-wave band
// elimination, the blocked dense elimination) is compiled.  True under hipcc and in the SIMT emulation build of
// tests/emul (CAVE_SIMT_EMUL: g++ with a shim <hip/hip_runtime.h> that runs every lane as a fiber), false in the
// single-lane serial test build.
#if defined(__HIPCC__) || defined(CAVE_SIMT_EMUL)
#define CAVE_GPU_CODE 1
#endif
// Ordering points that exist only because a wave's lanes run in lockstep and its LDS operations execute in issue
// order.  On the GPU they cost nothing (a compiler barrier) or an LDS-counter wait; the SIMT emulation turns them
// into a rendezvous of the wave / workgroup -- so a hand-over the source does not mark computes garbage there.
//   CAVE_WAVE_ORDER()   lanes of ONE wave exchange data through LDS across this point
//   CAVE_LDS_WAIT()     the same, plus the wave's own LDS operations have completed (one-wave workgroups)
//   CAVE_LDS_BARRIER()  workgroup barrier that waits for LDS traffic only (global loads stay in flight)
// streamed once-read data (the dense cones): non-temporal loads leave the caches to the data that is re-read
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL) && !defined(CAVE_NO_NT_LOADS)
#define CAVE_NT_LOAD_F4(p) ::cave::nt_load_f4(p)
#else
#define CAVE_NT_LOAD_F4(p) (*(p))
#endif
#if defined(CAVE_SIMT_EMUL)
#define CAVE_WAVE_ORDER() ::simt::wave_sync()
#define CAVE_LDS_WAIT() ::simt::wave_sync()
#define CAVE_LDS_BARRIER() ::simt::block_sync()
#elif defined(__HIPCC__)
#define CAVE_WAVE_ORDER() asm volatile("" ::: "memory")
#define CAVE_LDS_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define CAVE_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// Words two waves of a workgroup meet through (LDS): relaxed atomic accesses the compiler neither caches nor moves
// across the ordering points around them; a wave that polls one pauses between two looks (the SIMT emulation: the
// readfirstlane around the load is a rendezvous, which hands the processor to the other waves).
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL)
#define CAVE_FLAG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define CAVE_FLAG_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define CAVE_SPIN_PAUSE() __builtin_amdgcn_s_sleep(2)
#else
#define CAVE_FLAG_LOAD(p) (*(volatile int*)(p))
#define CAVE_FLAG_STORE(p, v) (*(volatile int*)(p) = (v))
#define CAVE_SPIN_PAUSE() do {} while (0)
#endif

// A loaded value the compiler must materialise HERE (it otherwise sinks a load into the branch that selects it, and
// a batch of independent LDS reads becomes a chain of round trips).  No code is emitted.
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL)
#define CAVE_PIN_F64(x) asm volatile("" : "+v"(x))
#else
#define CAVE_PIN_F64(x) do {} while (0)
#endif

// ---- optional phase timing (diagnostic builds only: -DCAVE_STAMPS; never in the shipped library).
// Cycle deltas are summed in per-wave registers (WaveCtx::st) and stored once per instance.
#if defined(CAVE_STAMPS) && defined(__HIPCC__)
#define CAVE_T0() unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define CAVE_ACC(slot) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    unsigned long long _t1 = __builtin_amdgcn_s_memtime(); \
    c.st[slot] += _t1 - _t0; _t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CAVE_T0() do {} while (0)
#define CAVE_ACC(slot) do {} while (0)
#endif

namespace cave {

// Address-space-typed pointers for the large-cone path.  Its arenas hand out generic pointers (an
// array may sit in LDS or in the global workspace), which compile to FLAT memory operations: those
// count on both wait counters, so every LDS access would also wait for outstanding HBM traffic.
// Code that knows where an array lives casts once and gets ds_* / global_* instructions.
template <class T, int SPACE>  // SPACE: 0 generic, 1 global, 3 LDS
struct SpacePtr { using type = T*; };
#if defined(__HIPCC__) && !defined(CAVE_SIMT_EMUL)
template <class T> struct SpacePtr<T, 1> { using type = __attribute__((address_space(1))) T*; };
template <class T> struct SpacePtr<T, 3> { using type = __attribute__((address_space(3))) T*; };
#endif
template <int SPACE, class T>
CAVE_HD typename SpacePtr<T, SPACE>::type space_cast(T* q) { return (typename SpacePtr<T, SPACE>::type)q; }

// Inside a REAL call the context, the view and the work record arrive by reference: their fields would be read through
// FLAT loads from the caller's private memory at every use.  A local copy lives in registers (or typed scratch); the
// context's mutable state (barrier parity, stamp counters) is handed back when the copy goes out of scope.
template <class C>
struct CtxLocal {
  C& ref;
  C c;
  CAVE_HD explicit CtxLocal(C& x) : ref(x), c(x) {}
  CAVE_HD ~CtxLocal() { ref = c; }
};

// for i = tid, tid + NT, ... < n:  op(i, ld(i)) -- with the loads of R iterations issued before the first op.  On the
// large-cone path the vectors live in the global workspace and a plain strided loop pays a full memory latency per
// iteration (seven for a 900-row vector on 128 threads; the stamp build of round 4 put ~150 such latencies into one
// Newton iteration of a 30 x 30 grid outside its band solve).  ld must be free of side effects (it is also called
// with the clamped index n - 1); per thread the order of the ops is that of the plain loop, so the bits are the same.
template <int R, int NT, class Ld, class Op>
CAVE_HD void strided_batched(int tid, int n, Ld&& ld, Op&& op) {
  if constexpr (R <= 1) {
    for (int i = tid; i < n; i += NT) op(i, ld(i));
  } else {
    for (int i0 = tid; i0 < n; i0 += R * NT) {
      decltype(ld(0)) vals[R];
#pragma unroll
      for (int u = 0; u < R; ++u) vals[u] = ld(i0 + u * NT < n ? i0 + u * NT : n - 1);
#pragma unroll
      for (int u = 0; u < R; ++u)
        if (i0 + u * NT < n) op(i0 + u * NT, vals[u]);
    }
  }
}
struct Ld2 { double a, b; };
struct Ld3 { double a, b, c; };
struct Ld4 { double a, b, c, d; };

// per-instance status codes (also in include/cave_hip.h)
enum : int32_t {
  ST_OK = 0,
  ST_NOT_CONVERGED = 1,  // Newton iteration cap hit (SciPy: RuntimeError, src/cave.py:307)
  ST_TOO_LARGE = 2,      // cone does not fit the LDS arena / nnz_cap / PMAX of this launch
  ST_BAD_INPUT = 3,      // non-finite input
};

// operating modes of the fused per-instance kernel
enum : int32_t {
  MODE_PROJECT = 0,    // _batch_project(..., 'nnls')            src/cave.py:231-264,298-309
  MODE_EXACT = 1,      // exactConeAlignedCosine fwd+bwd          src/cave.py:55-73,121-129
  MODE_INNER = 2,      // innerConeAlignedCosine QP branch (nnls) src/cave.py:206-219
  MODE_HEURISTIC = 3,  // innerConeAlignedCosine heuristic branch src/cave.py:201-204
  MODE_AVG = 4,        // _average_ctrs only                      src/cave.py:222-228
  MODE_IPM = 5,        // innerConeAlignedCosine, truncated interior-point iterate as the target (CaVE+ with an
                       // interior-point solver: src/cave.py:213-214,267-295; emulation, parity unpinned)
};

static constexpr uint32_t kTeamRow = 1024;  // entries: longer rows are summed by all waves of the workgroup together

// reference thresholds
static constexpr float kDropRowAbsSum = 1e-7f;  // src/cave.py:303
static constexpr float kAvgRowNorm = 1e-7f;     // src/cave.py:225
static constexpr double kNormClamp = 1e-8;      // src/cave.py:129,202,226 and F.cosine_similarity eps
static constexpr float kInsideRnorm = 1e-7f;    // src/cave.py:218

// row tags produced by classification
enum : uint8_t { ROW_DROP = 0, ROW_UNIT = 1, ROW_GENERAL = 2, ROW_AVG_VALID = 0x10 };

// Two-ended bump allocator over one LDS (or heap) buffer; all lanes run it uniformly.
// Persistent arrays grow from the bottom, build-phase temporaries from the top, so the
// temporaries can be dropped (release_top) before the solver's work arrays are carved.
struct Arena {
  unsigned char* base;
  uint32_t off;   // bottom watermark
  uint32_t top;   // top watermark (bytes [top, cap) are in use by temporaries)
  uint32_t cap;
  bool ovf;
  CAVE_HD void init(unsigned char* b, uint32_t c) { base = b; off = 0; cap = c & ~7u; top = cap; ovf = false; }
  template <class T>
  CAVE_HD T* get(uint32_t n) {
    uint32_t a = (off + 7u) & ~7u;
    uint64_t e = (uint64_t)a + (uint64_t)n * sizeof(T);
    if (e > top) { ovf = true; return reinterpret_cast<T*>(base); }
    off = (uint32_t)e;
    return reinterpret_cast<T*>(base + a);
  }
  template <class T>
  CAVE_HD T* get_top(uint32_t n) {
    uint64_t bytes = ((uint64_t)n * sizeof(T) + 7u) & ~7ull;
    if (bytes > top || top - bytes < off) { ovf = true; return reinterpret_cast<T*>(base); }
    top -= (uint32_t)bytes;
    return reinterpret_cast<T*>(base + top);
  }
  CAVE_HD void release_top() { top = cap; }
  CAVE_HD bool owns(const void* q) const {
    const unsigned char* u = (const unsigned char*)q;
    return u >= base && u < base + cap;
  }
  // like get(), but a request that does not fit returns null and leaves the arena untouched
  // (ALIGN: byte alignment of the result relative to the arena base, which is 16-byte aligned in LDS)
  template <class T, uint32_t ALIGN = 8u>
  CAVE_HD T* try_get(uint32_t n) {
    uint32_t a = (off + (ALIGN - 1u)) & ~(ALIGN - 1u);
    uint64_t e = (uint64_t)a + (uint64_t)n * sizeof(T);
    if (e > top) return nullptr;
    off = (uint32_t)e;
    return reinterpret_cast<T*>(base + a);
  }
};

// What the Newton solver needs to know about one cone (LDS-resident on the fast path; the large-cone
// path points into its global workspace or straight into the packed store).
struct SolveView {
  int d;                  // cost dimension
  int p;                  // reduced unknowns: one per general row (a +a/-a pair counts once)
  int n_valid;            // rows kept by the projection (0 -> empty cone -> proj = y)
  bool pm1;               // every reduced-row entry is +-1: sign in bit 15 of mcol / cvar, no value arrays
  const uint32_t* mptr;   // [p+1] CSR row pointers of the reduced rows
  const uint16_t* mcol;   // [nnzM]
  const float* mval;      // [nnzM]  (null when pm1)
  const uint8_t* vkind;   // [p]  1 = free multiplier (paired row), 0 = non-negative
  const uint32_t* cptr;   // [d+1] CSC of the reduced rows
  const uint16_t* cvar;   // [nnzM]
  const float* cvalc;     // [nnzM]  (null when pm1)
  const uint8_t* usign;   // [d]  bit0: +e_k row present, bit1: -e_k row present
  int nlong;              // reduced rows with more than kLongRow entries ...
  const uint32_t* longrow;  // ... and their indices
  double gcol_bound = 0.0;  // > 0: g = -M rc is summed COLUMN-wise in fixed point (cone_dense.h dense_gradient): (longest row) x (largest |entry|)
  int nteam = 0;            // of those, rows with more than kTeamRow entries (shared by the waves in the streamed gradient) ...
  const uint32_t* teamrow = nullptr;  // ... and their indices
  bool csc_far = false;   // "diet" layout (TSP-50 class: cone_instance.h run_packed_instance): cvar points into the packed
                          // store (global memory, signs in bit 15), read with batched loads; cptr stays in LDS
};

// packed lower triangle: H(i, j), j <= i, at i (i + 1) / 2 + j
CAVE_HD uint32_t tri_idx(uint32_t i, uint32_t j) {
  const uint32_t a = i > j ? i : j, b = i > j ? j : i;
  return a * (a + 1u) / 2u + b;
}

// Rows of H = M W M^T made on demand (one-wave band elimination of cones WITHOUT bound rows: grid shortest path).
// The band is then never materialised: no zeroing, no atomics, no 8 p (bw + 1) bytes written and read back per
// Newton iteration -- the elimination asks for a chunk of rows, one lane builds one row from the row's entries, the
// smoothed weights of their coordinates and the columns of those coordinates, in a fixed order (deterministic).
struct RbWork;
struct BandGen {
  bool on;
  const uint32_t* mptr;   // CSR of the reduced rows
  const uint16_t* mcol;
  const float* mval;      // null: entries are +-1, sign in bit 15 of mcol / cvar
  const uint32_t* cptr;   // CSC
  const uint16_t* cvar;
  const float* cvalc;
  const uint8_t* usign;
  const double* r;        // unclipped residual of the current iterate
  const struct RbWork* rb = nullptr;  // rows of the red-black Schur complement instead (cone_rb.h)
  double mu;              // 1 / (smoothing scale of this iteration), 0: binary weights
  double hdiag;           // bound on the diagonal of H (largest squared row norm): scale of the Levenberg shift
};

// Red-black reduction of the Newton systems of band cones without bound rows (cone_rb.h): an independent set of reduced
// rows ("red": no two of them share a coordinate -- every other node of a grid) is eliminated in closed form, the band
// LDL^T runs on the Schur complement of the others ("black").  All arrays live in the workspace block of the band,
// which such cones never materialise.
struct RbWork {
  bool on;
  int nB;          // black rows
  uint8_t* cls;    // [p] 1 = red, 2 = black
  uint16_t* pos;   // [p] black rows: position in the reduced system
  uint32_t* blk;   // [nB] reduced row at position q
  double* wt;      // [d] smoothed weight of each coordinate, this iteration
  double* hd;      // [p] diagonal of H + shift (red rows: its reciprocal, 0 = dropped)
  double* hdB;     // [nB] the same, black rows by position
  double* gB;      // [nB] reduced right-hand side
  uint32_t* rp;    // [nB + 1] recipe of row q of S: records rp[q] .. rp[q + 1]
  uint32_t* rec;   // two words per record (cone_rb.h)
  uint32_t* badj;  // [nB * 4] black rows by position: their (coordinate, red neighbour) pairs
  uint32_t* radj;  // [p * 4] red rows: their (coordinate, position of the black neighbour) pairs
};

// Dense reduced systems of the large-cone path (cone_dense.h): everything in LDS
struct DenseWork {
  bool on;         // this instance takes the dense path
  int nF, nI, ldS; // rows with free multipliers (eliminated first), rows with bounds, row stride of S
  double* A;       // [fold_entries(p)]  H, then its factor (int64 fixed point while being accumulated)
  double* dinv;    // [p]  reciprocal pivots of the eliminated rows
  double* z;       // [p]  right-hand side, eliminated alongside
  double* x;       // [p]  solution, in elimination order
  double* scr;     // [dense_scratch_entries(p)]  operands of one block step
  uint16_t* pos;   // [p]  position of reduced row i in the elimination order
  uint16_t* ord;   // [p]  reduced row at position q
  double* S;       // [nI * ldS]  Schur complement of the bound rows, both triangles
  double* sg;      // [nI]  reduced model gradient at the working point
  double* st;      // [nI]  working point (multipliers of the bound rows)
  double* ss;      // [nI]  step of one inner round
  double* sr;      // [nI]  right-hand side / S * step
  uint8_t* sact;   // [nI]  held at zero
  double hscale, hinv;  // fixed-point scale of the accumulation and its reciprocal
};

struct SolveWork {
  float* y;        // [d]
  double* res;     // [d]  residual y - M^T theta (unclipped while iterating, clipped on return)
  double* q;       // [d]  M^T (search direction)
  double* rc;      // [d]  Pi(res)
  double* theta;   // [p]
  double* ttry;    // [p]
  double* told;    // [p]  theta two iterations ago (zig-zag extrapolation)
  double* g;       // [p]
  double* dv;      // [p]  model gradient in the inner loop, then the search direction
  double* g2;      // [p]  right-hand side / H*step scratch
  double* step;    // [p]  Newton step of one inner round
  double* H;       // [p*ldh]  dense rows, or (band form) H[j*ldh + t] = H(j+t, j), t = 0..bw, ldh = bw+1
  bool tri = false;  // LDS-path general solver, diet layout: H is the packed lower triangle (tri_idx), p (p + 1) / 2 entries
  uint8_t* act;    // [p]
  float* wold;     // [d]  weight each coordinate currently has in H (fast path; the band path rebuilds H)
  const float* warm;  // [p] multipliers of an earlier solve of this cone (global memory), or null: starting point
  int ldh;
  // band form only (large-cone path, cone_band.h)
  bool band_hot;   // bwin, bz, step and act all live in LDS (typed fast variant of the band solver)
  bool band_wave;  // narrow band, everything hot: wave 0 runs solve_spd_band_wave (GPU build only)
  int bw;          // half bandwidth of M M^T in the reduced-row order
  double* bwin;    // [(bw+1)*(bw+1)] ring window of the rows being eliminated
  double* bfac;    // [p*(bw+1)] factor: bfac[k*ldh] = 1/d_k, bfac[k*ldh + t] = row k of the updated band
  double* bz;      // [p] right-hand side being eliminated
  double* bstg;    // [2*bch*(bw+1)] staging buffers for rows streamed from / to the workspace
  int bch;         // rows per staged chunk
  double hscale, hinv;  // fixed-point scale of the band Hessian's accumulation (large-cone path) and its reciprocal
  DenseWork dn;    // dense form (p <= bw + 1, p <= 128: TSP-100)
  BandGen gen;     // band rows on demand (band_wave, no bound rows: grid shortest path)
  RbWork rb;       // ... and their red-black reduction
  // lite solver with few bound rows (TSP-20: 20 free degree rows + <= 5 cut rows): partial elimination of the free
  // rows + active-set loop on the Schur complement of the bound rows (cone_core.h lite_model_step)
  bool ls_on;
  int ls_nF, ls_nI;
  double* ls_scr;  // LDS scratch: XS [p * nI], S [nI * (nI | 1)], four vectors of nI, flags
};

struct SolveResult {
  double f;        // 0.5*||res||^2
  int iters;
  int32_t status;
};

// 64-bit mix for row hashing (pair detection)
CAVE_HD uint64_t mix64(uint64_t h, uint64_t v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 32;
  return h;
}

CAVE_HD uint32_t f2u(float f) {
  union { float f; uint32_t u; } x;
  x.f = f;
  return x.u;
}

}  // namespace cave
