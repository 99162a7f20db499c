/* cave_hip.h — C ABI of the MI355X (gfx950) cone-projection backend for CaVE.
 *
 * This is the drop-in boundary: plain pointers and sizes, no
 * torch types.  Every pointer is a DEVICE pointer owned by the caller; inputs
 * are never written; all work is enqueued asynchronously on `stream`
 * (a hipStream_t passed as void*, NULL = default stream); nothing here
 * synchronises with the host.  Functions return CAVE_OK or a negative
 * CAVE_E_* code and never throw; per-instance solver outcomes are data
 * (`status[B]`, codes CAVE_ST_*), mirroring how the reference reports
 * Clarabel failures per instance (src/cave.py:293-294) while bad configuration
 * is rejected up front (src/cave.py:111-117,183-190).
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   cave_hip_cone_dense  mode PROJECT   _batch_project(..., solver='nnls') + _project_nnls
 *                                       src/cave.py:231-264, 298-309
 *                        mode EXACT     abstractConeAlignedCosine.forward + exact _get_projection
 *                                       src/cave.py:55-73, 121-129 (and its autograd backward)
 *                        mode INNER     innerConeAlignedCosine QP branch, nnls push-inside
 *                                       src/cave.py:206-219
 *                        mode HEURISTIC innerConeAlignedCosine heuristic branch  src/cave.py:201-204
 *                        mode AVG       _average_ctrs                            src/cave.py:222-228
 *   cave_hip_pack_*  / cave_hip_cone_packed
 *                                       optDatasetConstrs.ctrs storage + collate_fn padding
 *                                       src/dataset.py:72, 133-144 (device-resident replacement)
 *   cave_hip_cone_step                  one training step's worth of both on the dense wire format, in ONE launch:
 *                                       the forward / backward of the CURRENT batch (src/cave.py:55-73) from a
 *                                       transient store + the collate-side work on the NEXT batch the DataLoader
 *                                       has already produced (src/dataset.py:133-144)
 *   cave_hip_*_large                    the same three operators for cones whose reduced system does
 *                                       not fit registers / LDS (TSP-100, 30x30 shortest path: the
 *                                       reference runs them through the same _project_nnls,
 *                                       src/cave.py:298-309); caller-owned global workspace
 */
#ifndef CAVE_HIP_H
#define CAVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAVE_HIP_ABI_VERSION 10

/* return codes */
#define CAVE_OK 0
#define CAVE_E_INVALID (-1)  /* bad argument (null pointer, size out of range) */
#define CAVE_E_LAUNCH (-2)   /* HIP launch / attribute error, see cave_hip_last_error() */
#define CAVE_E_NO_DEVICE (-3)

/* per-instance status */
#define CAVE_ST_OK 0
#define CAVE_ST_NOT_CONVERGED 1 /* iteration cap reached (SciPy raises RuntimeError, src/cave.py:307) */
#define CAVE_ST_TOO_LARGE 2     /* cone did not fit this launch's LDS arena; retry with larger limits */
#define CAVE_ST_BAD_INPUT 3     /* non-finite values */

/* modes */
#define CAVE_MODE_PROJECT 0
#define CAVE_MODE_EXACT 1
#define CAVE_MODE_INNER 2
#define CAVE_MODE_HEURISTIC 3
#define CAVE_MODE_AVG 4
/* CaVE+ with a truncated interior-point projection (src/cave.py:213-214, 267-295: the reference runs Clarabel
 * with max_iter = 3 and uses the strictly interior iterate, normalised, as the target).  Here: `max_iter`
 * steps (<= 0: 3) of a primal path-following method on  min 1/2 ||y - A^T lam||^2 - tau * sum log lam_i  with
 * tau shrinking every step; the multipliers of the signed-unit rows are eliminated in closed form, which turns
 * their clip into its Chen-Harker-Kanzow-Smale smoothing.  Every multiplier of the iterate is > 0; proj / rnorm
 * are those of the iterate and tend to the exact projection as max_iter grows.  An emulation: Clarabel's own
 * iterates are not reproduced (no Clarabel in the image: parity unpinned).  Since v8 also on the large-cone path
 * (cave_hip_cone_dense_large / cave_hip_cone_packed_large): one band or dense LDL^T per interior-point step. */
#define CAVE_MODE_INNER_IPM 5

int32_t cave_hip_version(void);
/* thread-local, valid until the next failing call on this thread */
const char* cave_hip_last_error(void);
/* number of HIP devices visible (0 if none); does not create a context on failure */
int32_t cave_hip_device_count(void);

/* Launch limits.  nnz_cap: per-instance capacity for non-zero entries of the
 * dense block; lds_bytes: dynamic LDS per workgroup (<= 160 KiB).  Pass 0 for
 * either to let the library choose (cave_hip_default_limits reports the
 * choice so a caller can grow it after CAVE_ST_TOO_LARGE).
 * waves: wavefronts cooperating on one instance (workgroup = `waves` x 64 threads).
 *   0 = library default (2);  1 or 2: reduced systems up to 64 rows;  4: up to 32 rows;  8: four waves with
 *   the full register budget (up to 64 rows; for launches whose LDS arena leaves one workgroup per CU) (an instance with
 *   more reports CAVE_ST_TOO_LARGE: the host layer tries 4 first, then 2, and remembers per shape).
 * More waves shorten the per-instance critical path (useful while B is about the number of SIMDs,
 * 1024 on MI355X: TSP-20, B = 1024 takes 187 / 212 / 236 us with 4 / 2 / 1 waves); one wave per
 * instance maximises instances in flight (best throughput beyond ~1300 instances).
 * One workgroup per instance: B < 2^31. */
int32_t cave_hip_default_limits(int64_t m_max, int64_t d, int32_t* nnz_cap, int32_t* lds_bytes);

/* Fused per-instance operator on the reference's dense wire format.
 *   ctrs  [B, m_max, d] float32 row-major, zero-padded rows  (src/dataset.py:143)
 *   pred  [B, d]        float32; the kernel works on y = sign * pred
 *   mode  CAVE_MODE_*;  sign  -1 (EPO.MINIMIZE) / +1 (EPO.MAXIMIZE); for PROJECT pass +1
 *   inner_ratio  weight of the average normal (INNER, HEURISTIC)
 *   max_iter     Newton iteration cap (<=0: default 100); CAVE_MODE_INNER_IPM: interior-point steps (<=0: 3)
 * Outputs (any may be NULL):
 *   proj   [B, d]  projection of y onto cone{lam @ ctrs_b : lam >= 0}
 *   rnorm  [B]     ||y - proj||_2 (un-squared, nnls convention)
 *   target [B, d]  the constant target of the cosine loss (unit / pushed / heuristic; AVG: the average)
 *   loss   [B]     1 - cos(y, target)
 *   grad   [B, d]  d loss_b / d pred_b
 *   status [B]     CAVE_ST_*
 *   iters  [B]     Newton iterations used
 */
int32_t cave_hip_cone_dense(const float* ctrs, const float* pred, int64_t B, int64_t m_max, int64_t d,
                            int32_t mode, float sign, float inner_ratio, int32_t max_iter,
                            int32_t nnz_cap, int32_t lds_bytes, int32_t waves,
                            float* proj, float* rnorm, float* target, float* loss, float* grad,
                            int32_t* status, int32_t* iters, void* stream);

/* ---- device-resident packed cone store (replaces per-step dense padding) ---- */

/* Pass 1: per instance, count reduced rows and their non-zeros.
 *   n_rows [B], n_nnz [B], status [B] */
int32_t cave_hip_pack_count(const float* ctrs, int64_t B, int64_t m_max, int64_t d,
                            int32_t nnz_cap, int32_t lds_bytes, int32_t waves,
                            int32_t* n_rows, int32_t* n_nnz, int32_t* status, void* stream);

/* Packed store: structure-of-arrays, all device pointers, filled by cave_hip_pack_fill.
 *   row_off [n+1], nnz_off [n+1]  exclusive prefix sums of the pass-1 counts (int64)
 *   n_valid [n]        rows kept by the projection (0 = empty cone)
 *   flags   [n]        bit0: every reduced-row entry is +-1 (kernels then keep signs in the indices);
 *                      bit1 (set by the HOST on stores of the large-cone path, which reads them in place): ccol / cvar of
 *                      this instance already carry the sign in bit 15 (d, rows < 32768), cval / cvalc are not read
 *   usign   [n*d]      bit0: a +e_k row exists, bit1: a -e_k row exists
 *   avg     [n*d]      _average_ctrs of the instance (static, precomputed)
 *   vkind   [R]        1 = free multiplier (a +a/-a pair), 0 = non-negative     R = row_off[n]
 *   rlo,rhi [R]        CSR extent of each reduced row, relative to nnz_off[i]
 *   ccol, cval [Z]     CSR entries                                              Z = nnz_off[n]
 *   cptr    [n*(d+1)]  CSC column pointers, relative to nnz_off[i]
 *   cvar, cvalc [Z]    CSC entries (reduced-row index, value)
 */
typedef struct cave_cone_store {
  int64_t n;
  int32_t d;
  int32_t reserved;
  const int64_t* row_off;
  const int64_t* nnz_off;
  int32_t* n_valid;
  uint8_t* flags;
  uint8_t* usign;
  float* avg;
  uint8_t* vkind;
  uint32_t* rlo;
  uint32_t* rhi;
  uint16_t* ccol;
  float* cval;
  uint32_t* cptr;
  uint16_t* cvar;
  float* cvalc;
  /* Slot mode (both NULL in an exact-fit store).  When set, row_off / nnz_off only give each slot its
   * CAPACITY window (e.g. slot * max_rows, slot * max_nnz) and the actual sizes are n_rows[slot] / n_nnz[slot]:
   * cave_hip_pack_fill then needs no count pass (an instance that does not fit its window reports
   * CAVE_ST_TOO_LARGE and sets n_rows[slot] = -1, which cave_hip_cone_packed reports as CAVE_ST_TOO_LARGE too), which is how the dense operator runs small cones as
   * "pack into a transient slot store, then cave_hip_cone_packed" in one pass over the dense bytes. */
  int32_t* n_rows;
  int32_t* n_nnz;
  /* Warm start (both NULL: off).  warm_theta [R] float32, aligned with the reduced rows, holds the multipliers the
   * last converged projection of each instance ended with; warm_state [n] is 1 where that is valid (the caller
   * zeroes it to reset).  cave_hip_cone_packed(_large) then starts the Newton iteration there and refreshes both.
   * Cones are static per instance and predictions drift slowly during training (src/dataset.py:72); the
   * projection is unique, so results are the same as from a cold start (to the solver's tolerance). */
  float* warm_theta;
  uint8_t* warm_state;
  /* Red-black cache (v10; NULL: off).  rb_cache [n * rb_stride] bytes, ZEROED by the caller, rb_stride from
   * cave_hip_packed_large_rb_bytes(max_rows).  cave_hip_cone_packed_large keeps there, per instance, what its red-black
   * reduction of the band systems (grid shortest-path cones) derives from the STATIC cone alone -- the independent set
   * of reduced rows, the recipes of the Schur complement -- so that only the first projection of an instance builds
   * it (cones are static per instance, src/dataset.py:72).  Results do not depend on it. */
  uint8_t* rb_cache;
  int64_t rb_stride;
} cave_cone_store;

/* Pass 2: fill the store for instances [0, B) of `ctrs` at store slots [slot0, slot0+B). */
int32_t cave_hip_pack_fill(const float* ctrs, int64_t B, int64_t m_max, int64_t d,
                           int32_t nnz_cap, int32_t lds_bytes, int32_t waves,
                           const cave_cone_store* store, int64_t slot0, int32_t* status, void* stream);

/* Same operator as cave_hip_cone_dense, reading cones from the store:
 *   ids [B] int64 store slots (the collate_fn replacement hands these out). */
int32_t cave_hip_cone_packed(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                             int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                             int32_t waves, float* proj, float* rnorm, float* target, float* loss, float* grad,
                             int32_t* status, int32_t* iters, void* stream);

/* LDS bytes cave_hip_cone_packed needs for the largest instance of a store
 * (max_rows / max_nnz over instances, from the pass-1 counts).  all_pm1 != 0: every instance has
 * flags bit0 set, so no value arrays are staged (smaller arena -> more workgroups per CU).  all_pm1 == 1 also
 * reserves room for the index structures of the one-wave solver for small cones (d <= 256, <= 32 rows), which
 * launches of up to 2048 instances use; all_pm1 == 2: +-1 cones without that room (large batches: more
 * workgroups per CU matter more there); all_pm1 == 3 (v9): the "diet" layout of +-1 cones with more than 32 reduced
 * rows (TSP-50: 107 KB -> 76 KB, two workgroups per compute unit instead of one): H as a packed lower triangle, the
 * CSC entries and the average normal read in place from the store.  It is taken, per instance, by
 * cave_hip_cone_packed(waves = 8) when the instance's ordinary arena exceeds the lds_bytes of the launch and the
 * store carries the signs in its indices (flags bit 1). */
int32_t cave_hip_packed_lds_bytes(int64_t d, int32_t max_rows, int32_t max_nnz, int32_t all_pm1);

/* ------------------------------------------------------------------ large-cone path
 * Same operators, same per-instance semantics and outputs, for cones beyond the fast path's limits
 * (more than 64 reduced rows, or more non-zeros than 160 KiB of LDS holds).  Persistent workgroups
 * (4 waves; cave_hip_cone_packed_large takes 2-wave workgroups when the batch needs more than two
 * workgroups per CU and four fit the LDS); each works in its own slice of a caller-owned, 16-byte aligned device `workspace` of
 * n_slots * slice_bytes bytes (n_slots = number of workgroups launched, at most B are used; a few per
 * CU is enough).  The Newton systems are kept as symmetric bands (band = reduced rows x (half
 * bandwidth + 1) entries) and solved by an LDL^T band elimination (narrow bands -- half bandwidth
 * 4 .. 34, grid shortest-path cones -- four pivots per step on one or two waves).  An instance that does not fit its
 * slice reports CAVE_ST_TOO_LARGE: retry with a larger slice.
 *   nnz_cap       non-zeros kept per instance
 *   band_entries  expected rows x (bandwidth + 1) of the reduced system (sizing hint)
 *   lds_bytes     LDS per workgroup used for the small hot arrays, 0 = 64 KiB */
int64_t cave_hip_large_slice_bytes(int64_t m_max, int64_t d, int64_t nnz_cap, int64_t band_entries);
int64_t cave_hip_packed_large_slice_bytes(int64_t d, int64_t max_rows, int64_t band_entries);
/* LDS per workgroup that keeps the hot arrays of the band solver (ring window, staging, two row vectors, flags)
 * on chip for reduced systems of up to max_rows rows and half bandwidth max_bw (v7).  Narrow bands
 * (max_bw <= 34) are eliminated by ONE wave per instance; the figure returned for them is the exact need, so that
 * four workgroups share a CU (the 2- and 1-wave forms of cave_hip_cone_packed_large are chosen from it). */
int32_t cave_hip_packed_large_lds_bytes(int32_t max_rows, int32_t max_bw);
/* Bytes per instance of cave_cone_store.rb_cache for reduced systems of up to max_rows rows (v10); 0: such cones never
 * take the red-black reduction. */
int64_t cave_hip_packed_large_rb_bytes(int64_t max_rows);

int32_t cave_hip_cone_dense_large(const float* ctrs, const float* pred, int64_t B, int64_t m_max, int64_t d,
                                  int32_t mode, float sign, float inner_ratio, int32_t max_iter, int64_t nnz_cap,
                                  int32_t lds_bytes, void* workspace, int64_t slice_bytes, int32_t n_slots, float* proj,
                                  float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                                  int32_t* iters, void* stream);

/* store == NULL: count pass (n_rows / n_nnz per instance, as cave_hip_pack_count);
 * store != NULL: fill pass into slots [slot0, slot0 + B) (as cave_hip_pack_fill; n_rows / n_nnz unused). */
int32_t cave_hip_pack_large(const float* ctrs, int64_t B, int64_t m_max, int64_t d, int64_t nnz_cap, void* workspace,
                            int64_t slice_bytes, int32_t n_slots, int32_t* n_rows, int32_t* n_nnz,
                            const cave_cone_store* store, int64_t slot0, int32_t* status, void* stream);

/* reads the cones in place from the store; the workspace only holds the solver's work arrays.
 *   waves (v8)  wavefronts per workgroup: 1, 2 or 4; 0 = library default (4, or 2 when the batch needs more than
 *               two workgroups per compute unit and four workgroups fit the LDS).  The library reads no
 *               environment variable and keeps no state between calls. */
int32_t cave_hip_cone_packed_large(const cave_cone_store* store, const int64_t* ids, const float* pred, int64_t B,
                                   int32_t mode, float sign, float inner_ratio, int32_t max_iter, int32_t lds_bytes,
                                   int32_t waves, void* workspace, int64_t slice_bytes, int32_t n_slots, float* proj,
                                   float* rnorm, float* target, float* loss, float* grad, int32_t* status,
                                   int32_t* iters, void* stream);

/* ------------------------------------------------------------------ fused step (v9)
 * Small +-1 cones on the dense wire format (TSP-20, small grids: d <= 256, <= 32 reduced rows in the order
 * [free | <= 8 bound rows], <= 8 entries per column, <= 1536 non-zeros -- what the one-wave solver takes).
 *
 * A step on the dense format is  (a) stream the dense block + build the reduced cone  ->  (b) Newton solve + loss +
 * gradient.  (a) depends on the cones only, and the training loop has the cones of batch i+1 before it has the
 * prediction of batch i+1 (the DataLoader collates ahead), so cave_hip_cone_step runs (b) of the CURRENT batch and (a)
 * of the NEXT batch in one grid: B one-wave solve instances first in the block order, then B_next four-wave pack
 * workgroups, which fill what the solve waves leave of each compute unit.  One stream, no events.
 *
 * cave_lite_store: transient per-batch store in the layout the one-wave solver reads (caller-owned device arrays):
 *   hdr    [n*8]    int32: state (1 = holds a cone, -1 = the cone does not qualify, 0 = never packed), reduced rows p,
 *                   non-zeros, free rows nF (rows [0, nF) have free multipliers), rows kept by the projection,
 *                   longest column, CSR entries per lane (8 / 16 / 24), spare
 *   usign  [n*d]    bit0: a +e_k row exists, bit1: a -e_k row exists
 *   avg    [n*d]    _average_ctrs of the instance
 *   rowptr [n*33]   CSR row pointers of the reduced rows
 *   ell    [n*4*d]  uint32: the column of coordinate k as eight 16-bit entries (reduced row | sign << 15; unused: row 32)
 *   csr16  [n*768]  uint32: CSR entries as 16-bit words (column | sign << 15 | row-end marks), lane-major groups of 8
 *   rl     [n*32]   lane holding the last entry of reduced row i
 * (ell / csr16 need 16-byte aligned bases.) */
typedef struct cave_lite_store {
  int64_t n;
  int32_t d;
  int32_t reserved;
  int32_t* hdr;
  uint8_t* usign;
  float* avg;
  uint32_t* rowptr;
  uint32_t* ell;
  uint32_t* csr16;
  uint8_t* rl;
} cave_lite_store;

/* dynamic LDS per workgroup of cave_hip_cone_step for dense batches of shape (m_max, d); <= 0: the shape does not
 * qualify (d > 256, or five workgroups would not fit a compute unit: use the general operators) */
int32_t cave_hip_step_lds_bytes(int64_t m_max, int64_t d);

/* Solve half: B instances of `solve` -- slot ids[b], or slot b when ids is NULL (a transient per-batch store packed
 *   by an earlier call) -- with predictions pred [B, d]; outputs as cave_hip_cone_dense (modes PROJECT / EXACT / INNER /
 *   HEURISTIC / AVG; an instance whose slot holds no cone reports CAVE_ST_TOO_LARGE, a slot out of range
 *   CAVE_ST_BAD_INPUT).  B = 0 (solve may be NULL): pack only.
 * Pack half: instances [0, B_next) of next_ctrs [B_next, m_max, d] into slots [0, B_next) of `next` (a different
 *   store than `solve`); pack_status [B_next] or NULL.  B_next = 0 (next_ctrs / next may be NULL): solve only.
 * cu_tickets: 4096 uint32 of device memory, zeroed once by the caller, shared by the launches of one device (per
 *   compute-unit counters that spread the solve waves over the SIMDs; the library keeps no state of its own). */
#define CAVE_STEP_ZERO_FAILED 1 /* flags: an instance whose status is not CAVE_ST_OK gets loss 0 and a zero gradient (a
                                * training loop that examines `status` a step later must not feed NaN to its optimizer) */
int32_t cave_hip_cone_step(const cave_lite_store* solve, const int64_t* ids, const float* pred, int64_t B, int32_t mode, float sign,
                           float inner_ratio, int32_t max_iter, int32_t flags, float* proj, float* rnorm, float* target, float* loss,
                           float* grad, int32_t* status, int32_t* iters, const float* next_ctrs, int64_t B_next,
                           int64_t m_max, int64_t d, const cave_lite_store* next, int32_t* pack_status,
                           uint32_t* cu_tickets, void* stream);

/* Device-resident stores: cones are static per instance (src/dataset.py:72), so a packed store whose cones qualify
 * builds the lite slots of ALL its instances once (slot i of `dst` from slot i of `src`, dst->n >= src->n) and then
 * serves batches of ids through the solve half of cave_hip_cone_step (no pack half).  status [src->n] or NULL:
 * CAVE_ST_OK, or CAVE_ST_TOO_LARGE for a cone the one-wave solver does not take (its slot is marked so). */
int32_t cave_hip_lite_from_packed(const cave_cone_store* src, const cave_lite_store* dst, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAVE_HIP_H */
