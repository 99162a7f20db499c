"""Batched cone projection on MI355X — the `solver='hip'` arm.

Host-side mirror of the reference's plug point: a lazily imported function
``f(tight_ctrs, signed_cost, **solver_kwargs) -> (proj (B,d), rnorm (B,))``
living next to ``project_apgd`` (/root/reference src/qpsolver.py:11-63, called
from ``_batch_project`` src/cave.py:242-244), but with **nnls semantics**
(src/cave.py:298-309): exact Euclidean projection onto
``cone{lam @ ctrs_b : lam >= 0}``, zero-padded rows ignored, empty cone returns
the input, ``rnorm`` is the un-squared residual norm.

One call is one HIP kernel launch (cave_amd/csrc/kernels.h), or two for small cones on the dense wire
format (the "split" form below: a 4-wave streaming pack kernel, then a 1-wave solve kernel);
this module only marshals pointers, picks launch limits and turns per-instance
status codes into the reference's error behaviour.
"""

from __future__ import annotations

import torch

from . import _lib
from ._lib import (MODE_AVG, MODE_EXACT, MODE_HEURISTIC, MODE_INNER, MODE_PROJECT, ST_BAD_INPUT,
                   ST_NOT_CONVERGED, ST_OK, ST_TOO_LARGE)

__all__ = ["project_hip", "average_ctrs_hip", "cone_op_dense", "HipSolverError", "PreparedCones", "prepare_dense",
           "cone_op_prepared", "step_lds_bytes"]


class HipSolverError(RuntimeError):
    """Per-instance solver failure (SciPy's nnls raises RuntimeError on its iteration cap, src/cave.py:307)."""


def _as_device(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _grow_limits(m: int, d: int) -> tuple[int, int]:
    """Largest arena a workgroup can have (160 KiB) and a non-zero capacity that leaves room for the
    rest: the scan output (8 B per non-zero) is a build-phase temporary, so it may take about half."""
    cap = int(_lib.MAX_LDS * 0.55) // 8 - 256
    return int(min(cap, max(m * d, 64))), _lib.MAX_LDS


# shapes (m_max, d) whose cones did not fit the default (structured-cone) arena: go straight to the
# tier that worked next time instead of paying failed launches per call
#   tier 1: full 160 KiB LDS arena (one workgroup per CU, four waves with the wide register budget: waves=8);
#   tier 2: large-cone path (global workspace)
_tier: dict[tuple[int, int], int] = {}
# shapes (m_max, d) whose cones were seen (by a status-checked launch) to fit / not to fit the 4-wave
# workgroup shape (reduced systems up to 32 rows), the fastest one while the GPU has idle SIMDs
_wide_ok: dict[tuple[int, int], bool] = {}


def _auto_waves(B: int, m: int, d: int, check: bool) -> int:
    """Waves per instance of the FUSED dense kernel when the caller leaves it open (cones with d <= 256 normally
    take the split form instead, see launch_split).  Four waves shorten the streaming scan while the GPU has idle
    SIMDs (TSP-20, B = 1024, rotating batches, r02: 219 us with 4 waves, 193 with 2, 196 with 1); one wave per
    instance wins beyond ~1300 instances (more instances in flight).  Four waves hold reduced systems up to 32
    rows and are only used once a status-checked launch has shown the shape fits."""
    if B > 1280:
        return 1  # reduced systems up to 64 rows, like the 2-wave shape
    ok = _wide_ok.get((m, d))
    if ok is True or (ok is None and check):
        return 4
    return 2

_large_hint: dict[tuple[int, int], tuple[int, int]] = {}  # (m, d) -> (nnz_cap, band_entries) that fitted
# shapes that have completed a clean status-checked call (any wave count, any tier): only these may be
# launched unchecked / lazily with what was learnt; `forget_shape` takes a shape out again
_settled: set[tuple[int, int]] = set()


def forget_shape(m: int, d: int) -> None:
    """Drop everything remembered about (m_max, d).  Called when an unchecked or lazily checked launch
    reported CAVE_ST_TOO_LARGE: the caches are keyed by shape only, and a later batch of the same shape
    may hold bigger cones (more reduced rows / non-zeros) than the one that settled it.  The next call
    for the shape then runs status-checked and walks the tiers again."""
    key = (int(m), int(d))
    _tier.pop(key, None)
    _wide_ok.pop(key, None)
    _large_hint.pop(key, None)
    _split_ok.pop(key, None)
    _step_ok.pop(key, None)
    _settled.discard(key)


def _large_guess(m: int, d: int) -> tuple[int, int]:
    """Initial (nnz_cap, band_entries) of the large-cone path: structured cones carry <= d unit entries
    plus a few sparse rows; the reduced systems of the CaVE benchmarks are ~sqrt(2d) dense rows (TSP
    degree rows) or ~d/2 rows of bandwidth ~sqrt(d/2) (grid flow rows): both well under 32*d entries."""
    cap = int(min(max(m * d, 64), 4 * (m + d) + 256))
    return cap, 32 * d + 4096


def fast_path_cannot_fit(d: int) -> bool:
    """The per-coordinate arrays of the LDS-resident solver (sign byte, column pointers, y, avg,
    residual, clipped residual, direction, flags: ~38 bytes per cost coordinate) alone exceed 160 KiB."""
    return 38 * d + 4096 > _lib.MAX_LDS


# ---- "split" form of the dense operator for small cones (d <= 256):
#   kernel 1  cave_hip_pack_fill in slot mode: four waves per instance stream the dense block and leave the
#             reduced cone in a fixed-capacity slot of a transient device store (one pass over the dense bytes);
#   kernel 2  cave_hip_cone_packed with ONE wave per instance, whose small-cone ("lite") solver wants ~200
#             registers -- more than the 4-wave streaming shape can give without losing residency.
# The two kernels each get the launch shape they need; the transient store costs ~10 KB of extra HBM traffic per
# instance against 178 KB of dense input (TSP-20).  Instances beyond the slot capacity (more than 32 reduced
# rows / 1536 non-zeros, or entries other than +-1) report TOO_LARGE and the batch falls back to the fused kernel.
SPLIT_MAX_D, SPLIT_ROWS, SPLIT_NNZ = 256, 32, 1536
_slot_stores: dict = {}
_split_ok: dict[tuple[int, int], bool] = {}  # (m, d) -> the split form fitted every instance of a checked batch


class _SlotStore:
    def __init__(self, dev, B: int, d: int):
        import ctypes as C

        self.B, self.d = B, d
        t = {}
        ar = torch.arange(B + 1, dtype=torch.int64, device=dev)
        t["row_off"], t["nnz_off"] = ar * SPLIT_ROWS, ar * SPLIT_NNZ
        R, Z = B * SPLIT_ROWS, B * SPLIT_NNZ
        t["n_valid"] = torch.zeros(B, dtype=torch.int32, device=dev)
        t["flags"] = torch.zeros(B, dtype=torch.uint8, device=dev)
        t["usign"] = torch.zeros(B * d, dtype=torch.uint8, device=dev)
        t["avg"] = torch.zeros(B * d, dtype=torch.float32, device=dev)
        t["vkind"] = torch.zeros(R, dtype=torch.uint8, device=dev)
        t["rlo"] = torch.zeros(R, dtype=torch.int32, device=dev)
        t["rhi"] = torch.zeros(R, dtype=torch.int32, device=dev)
        t["ccol"] = torch.zeros(Z, dtype=torch.int16, device=dev)
        t["cval"] = torch.zeros(Z, dtype=torch.float32, device=dev)
        t["cptr"] = torch.zeros(B * (d + 1), dtype=torch.int32, device=dev)
        t["cvar"] = torch.zeros(Z, dtype=torch.int16, device=dev)
        t["cvalc"] = torch.zeros(Z, dtype=torch.float32, device=dev)
        t["n_rows"] = torch.zeros(B, dtype=torch.int32, device=dev)
        t["n_nnz"] = torch.zeros(B, dtype=torch.int32, device=dev)
        self.t = t
        self.c = _lib.Store(n=B, d=d, reserved=0, **{k: v.data_ptr() for k, v in t.items()})
        self.ref = C.byref(self.c)
        self.pack_status = torch.empty(B, dtype=torch.int32, device=dev)
        self.lds_bytes = int(_lib.load_library().cave_hip_packed_lds_bytes(d, SPLIT_ROWS, SPLIT_NNZ, 1))
        self.gen = 0  # bumped every time prepare_dense hands this store out (PreparedCones.stale)


def _slot_store(dev, B: int, d: int) -> _SlotStore:
    key = (dev, B, d)
    st = _slot_stores.get(key)
    if st is None:
        if len(_slot_stores) >= 4:
            _slot_stores.pop(next(iter(_slot_stores)))
        st = _slot_stores[key] = _SlotStore(dev, B, d)
    return st


# ---- prepared form: the two stages of a step on the dense format, decoupled and then FUSED ACROSS STEPS.
# The pack stage (stream the dense block, build the reduced cone) depends on the cones only, not on the prediction,
# and a training loop knows the cones of the NEXT batch before it has the next prediction (the DataLoader has
# already collated it: src/dataset.py:133-144).  `prepare_dense(ctrs)` runs the pack stage of a batch into a transient
# "lite store"; `prep.then(next_ctrs)` attaches the following batch, and `cone_op_prepared(prep, pred, ...)` then
# launches ONE grid (cave_hip_cone_step): the one-wave solve blocks of this batch first, the four-wave pack blocks of
# the next batch behind them, filling what the solve waves leave of each compute unit.  Steady state per step =
# max(pack, solve) instead of their sum, on one stream: no side stream, no event, no spacer kernel (rounds 2-3 ran the
# pack on a side stream behind a tuned `torch.cuda._sleep`, which decided a dispatch race and depended on how the
# runtime maps streams to hardware queues).  `cave_amd.dataset.prefetch(loader)` does the wiring for a DataLoader.
STEP_MAX_B = 2048   # the one-wave solver is a latency design: beyond ~2 instances per SIMD the general path wins
STEP_POOL = 3       # lite stores cycled per (device, stream, B, d); two are in use by a running chain
_step_pool: dict = {}
_step_lds: dict = {}
_step_ok: dict[tuple[int, int], bool] = {}  # (m, d) -> a checked batch of this shape had a cone the lite form does not take
_tickets: dict = {}


class _LiteSlots:
    """B slots of a cave_lite_store (include/cave_hip.h) + the pack status of the batch it holds (or, for a
    device-resident ConeStore, of all its instances)."""

    def __init__(self, dev, B: int, d: int):
        import ctypes as C

        self.B, self.d = B, d
        t = {
            "hdr": torch.zeros(B * 8, dtype=torch.int32, device=dev),
            "usign": torch.zeros(B * d, dtype=torch.uint8, device=dev),
            "avg": torch.zeros(B * d, dtype=torch.float32, device=dev),
            "rowptr": torch.zeros(B * 33, dtype=torch.int32, device=dev),
            "ell": torch.zeros(B * 4 * d, dtype=torch.int32, device=dev),
            "csr16": torch.zeros(B * 768, dtype=torch.int32, device=dev),
            "rl": torch.zeros(B * 32, dtype=torch.uint8, device=dev),
        }
        self.t = t
        self.c = _lib.LiteStore(n=B, d=d, reserved=0, **{k: v.data_ptr() for k, v in t.items()})
        self.ref = C.byref(self.c)
        self.pack_status = torch.empty(B, dtype=torch.int32, device=dev)
        self.gen = 0  # bumped every time the store is handed out (PreparedCones.stale)


def _tickets_for(dev) -> torch.Tensor:
    t = _tickets.get(dev)
    if t is None:
        t = _tickets[dev] = torch.zeros(4096, dtype=torch.int32, device=dev)
    return t


def step_lds_bytes(m: int, d: int) -> int:
    """LDS per workgroup of the fused step kernel for dense batches of shape (m_max, d); <= 0: does not qualify."""
    key = (int(m), int(d))
    v = _step_lds.get(key)
    if v is None:
        v = _step_lds[key] = int(_lib.load_library().cave_hip_step_lds_bytes(key[0], key[1]))
    return v


def _take_store(dev, B: int, d: int, avoid=None) -> _LiteSlots:
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, B, d)
    pool = _step_pool.get(key)
    if pool is None:
        if len(_step_pool) >= 8:
            _step_pool.pop(next(iter(_step_pool)))
        pool = _step_pool[key] = []
    if len(pool) < STEP_POOL:
        ss = _LiteSlots(dev, B, d)
    else:
        ss = pool.pop(0)  # round robin: the store handed out STEP_POOL calls ago (its holder is stale from now on)
        if ss is avoid:
            pool.append(ss)
            ss = pool.pop(0)
    pool.append(ss)
    ss.gen += 1
    return ss


class PreparedCones:
    """A dense (B, m_max, d) batch whose reduced cones sit in a transient lite store (packed by an earlier launch on
    the same stream).  Usable in place of `tight_ctrs` in a loss call.  `then(next_ctrs)` attaches the batch that
    follows: the loss call then packs it in the same launch and leaves its PreparedCones in `.next`.  Keeps the dense
    tensor alive for the fallback of a batch the lite form does not take -- or whose store has been handed out again
    in the meantime (`gen` no longer matches: more prepared batches held than the pool has stores)."""

    def __init__(self, ctrs: torch.Tensor, store: _LiteSlots, gen: int):
        self.ctrs, self.store, self.gen = ctrs, store, gen
        self.shape = tuple(ctrs.shape)
        self.follow = None   # dense tensor of the batch after this one (consumed by the first solve)
        self.next = None     # what to pass for that batch: a PreparedCones, or the tensor itself

    def stale(self) -> bool:
        return self.gen != self.store.gen

    def then(self, next_ctrs: "torch.Tensor | None") -> "PreparedCones":
        self.follow, self.next = next_ctrs, None
        return self

    # a training loop moves every field of a batch to the device (code_sample.py:50): nothing to move here
    def cuda(self, *a, **k) -> "PreparedCones":
        return self

    def to(self, *a, **k) -> "PreparedCones":
        return self

    @property
    def device(self):
        return self.ctrs.device


def _step_qualifies(t) -> bool:
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 3):
        return False
    B, m, d = t.shape
    return 0 < m and d <= SPLIT_MAX_D and 0 < B <= STEP_MAX_B and _step_ok.get((m, d)) is not False and step_lds_bytes(m, d) > 0


def _launch_step(solve, pred, B, mode, sign, inner_ratio, max_iter, out, status, iters, nxt_ctrs, nxt_store, ids=None,
                 zero_failed=False):
    lib = _lib.load()
    Bn, mn, dn = (nxt_ctrs.shape if nxt_ctrs is not None else (0, 0, solve.d))
    dev = status.device if status is not None else nxt_ctrs.device
    rc = lib.cave_hip_cone_step(
        solve.ref if solve is not None else None, _lib.ptr(ids), _lib.ptr(pred), B, int(mode), float(sign), float(inner_ratio),
        int(max_iter), 1 if zero_failed else 0,
        _lib.ptr(out.get("proj")), _lib.ptr(out.get("rnorm")), _lib.ptr(out.get("target")), _lib.ptr(out.get("loss")),
        _lib.ptr(out.get("grad")), _lib.ptr(status), _lib.ptr(iters),
        _lib.ptr(nxt_ctrs), Bn, mn, dn, nxt_store.ref if nxt_store is not None else None,
        _lib.ptr(nxt_store.pack_status) if nxt_store is not None else None, _lib.ptr(_tickets_for(dev)), _lib.current_stream())
    _lib.check(rc, "cave_hip_cone_step")


def prepare_dense(tight_ctrs: torch.Tensor) -> "PreparedCones | torch.Tensor":
    """Run the pack stage of a dense batch now, on the current stream (a pack-only launch of the step kernel).
    Returns the tensor itself when the shape does not qualify (the loss call then takes the ordinary path)."""
    _lib.load()
    if not _step_qualifies(tight_ctrs):
        return tight_ctrs
    dev = tight_ctrs.device
    ctrs = _as_device(tight_ctrs, dev)
    B, m, d = ctrs.shape
    with torch.cuda.device(dev):
        ss = _take_store(dev, B, d)
        _launch_step(None, None, 0, MODE_PROJECT, 1.0, 0.0, 0, {}, None, None, ctrs, ss)
    return PreparedCones(ctrs, ss, ss.gen)


def cone_op_prepared(prep: PreparedCones, pred_cost: torch.Tensor, mode: int, sign: float = 1.0, inner_ratio: float = 0.2, *,
                     max_iter: int = 0, check: bool = True, zero_failed: bool = False,
                     outputs: tuple[str, ...] = ("proj", "rnorm")) -> dict[str, torch.Tensor]:
    """The solve stage for a prepared batch (same outputs as cone_op_dense) and, in the same launch, the pack stage of
    the batch attached with `prep.then(...)`, whose PreparedCones is left in `prep.next`.  A batch with a cone the lite
    form does not take falls back to cone_op_dense on the dense tensor (checked calls only; unchecked calls report
    CAVE_ST_TOO_LARGE in `status`)."""
    _lib.load()
    B, m, d = prep.shape
    dev = prep.ctrs.device
    follow, prep.follow = prep.follow, None
    if prep.stale() or mode == _lib.MODE_INNER_IPM:
        # its store now holds a later batch (or the mode is not one of the step kernel's): solve from the dense tensor
        if follow is not None:
            prep.next = prepare_dense(follow)
        return cone_op_dense(prep.ctrs, pred_cost, mode, sign, inner_ratio, max_iter=max_iter, check=check, outputs=outputs)
    pred = _as_device(pred_cost, dev)
    if pred.shape != (B, d):
        raise ValueError(f"pred_cost must have shape ({B}, {d}), got {tuple(pred.shape)}")
    out: dict[str, torch.Tensor] = {}
    with torch.cuda.device(dev):
        for name in outputs:
            out[name] = torch.empty((B,) if name in ("rnorm", "loss") else (B, d), dtype=torch.float32, device=dev)
        status = torch.empty(B, dtype=torch.int32, device=dev)
        iters = torch.empty(B, dtype=torch.int32, device=dev)
        out["status"], out["iters"] = status, iters
        nctrs = nstore = None
        if follow is not None:
            if _step_qualifies(follow):
                nctrs = _as_device(follow, dev)
                nstore = _take_store(dev, int(nctrs.shape[0]), int(nctrs.shape[2]), avoid=prep.store)
                prep.next = PreparedCones(nctrs, nstore, nstore.gen)
            else:
                prep.next = follow
        _launch_step(prep.store, pred, B, mode, sign, inner_ratio, max_iter, out, status, iters, nctrs, nstore,
                     zero_failed=zero_failed)
        if zero_failed:
            out["zero_failed"] = True  # (the kernel wrote loss 0 / gradient 0 for instances whose status is not OK)
        if check:
            if bool((status == ST_TOO_LARGE).any()):
                _step_ok[(m, d)] = False
                return cone_op_dense(prep.ctrs, pred_cost, mode, sign, inner_ratio, max_iter=max_iter, check=True,
                                     outputs=outputs)
            _raise_for_status(status, "solver='hip' (prepared)")
    return out


def _raise_for_status(status: torch.Tensor, what: str) -> None:
    st = status.cpu()
    if bool((st == ST_OK).all()):
        return
    bad = int((st != ST_OK).sum())
    first = int((st != ST_OK).nonzero()[0])
    code = int(st[first])
    if code == ST_NOT_CONVERGED:
        raise HipSolverError(f"{what}: Maximum number of iterations reached ({bad} instance(s), first index {first}).")
    if code == ST_TOO_LARGE:
        raise HipSolverError(
            f"{what}: {bad} cone(s) (first index {first}) did not fit the workspace of the large-cone path "
            "after four 4x size increases (non-zeros / band of the reduced system).")
    if code == ST_BAD_INPUT:
        raise ValueError(f"{what}: non-finite input in {bad} instance(s), first index {first}.")
    raise HipSolverError(f"{what}: unknown status {code}")


def cone_op_dense(tight_ctrs: torch.Tensor, pred_cost: torch.Tensor | None, mode: int, sign: float = 1.0,
                  inner_ratio: float = 0.2, *, max_iter: int = 0, nnz_cap: int = 0, lds_bytes: int = 0,
                  waves: int = 0, check: bool = True, outputs: tuple[str, ...] = ("proj", "rnorm")) -> dict[str, torch.Tensor]:
    """Run the fused per-instance kernel on the reference's dense wire format.

    tight_ctrs (B, m_max, d) float32 zero-padded (src/dataset.py:143); pred_cost (B, d).
    ``outputs`` selects which of proj / rnorm / target / loss / grad are materialised.
    ``waves``: wavefronts cooperating per instance (0 = chosen from the batch size and what is known
    about the shape, see ``_auto_waves``; 1 or 2: reduced systems up to 64 rows; 4: up to 32 rows).
    With ``check=True`` (default) the per-instance status is read back (one host sync):
    a batch with a cone that does not fit is retried with one wave per instance and the largest
    LDS arena, then on the large-cone path (global workspace, band Newton systems: TSP-100,
    30x30 grids); the tier that worked is remembered per (m_max, d).  Anything else raises.
    """
    lib = _lib.load()
    if tight_ctrs.dim() != 3:
        raise ValueError("tight_ctrs must have shape (B, m_max, d)")
    B, m, d = tight_ctrs.shape
    dev = tight_ctrs.device if tight_ctrs.is_cuda else (
        pred_cost.device if pred_cost is not None and pred_cost.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    ctrs = _as_device(tight_ctrs, dev)
    pred = None
    if pred_cost is not None:
        if pred_cost.shape != (B, d):
            raise ValueError(f"pred_cost must have shape ({B}, {d}), got {tuple(pred_cost.shape)}")
        pred = _as_device(pred_cost, dev)
    elif mode != MODE_AVG:
        raise ValueError("pred_cost is required")
    out: dict[str, torch.Tensor] = {}
    with torch.cuda.device(dev):
        for name in outputs:
            shape = (B,) if name in ("rnorm", "loss") else (B, d)
            out[name] = torch.empty(shape, dtype=torch.float32, device=dev)
        status = torch.empty(B, dtype=torch.int32, device=dev)
        iters = torch.empty(B, dtype=torch.int32, device=dev)
        out["status"], out["iters"] = status, iters
        if B == 0:
            return out

        def launch(cap: int, lds: int, nw: int) -> None:
            rc = lib.cave_hip_cone_dense(
                _lib.ptr(ctrs), _lib.ptr(pred), B, m, d, int(mode), float(sign), float(inner_ratio),
                int(max_iter), int(cap), int(lds), int(nw),
                _lib.ptr(out.get("proj")), _lib.ptr(out.get("rnorm")), _lib.ptr(out.get("target")),
                _lib.ptr(out.get("loss")), _lib.ptr(out.get("grad")), _lib.ptr(status), _lib.ptr(iters),
                _lib.current_stream())
            _lib.check(rc, "cave_hip_cone_dense")

        # LDS of the large path's hot arrays (band window + staging).  The band is only known inside the
        # kernel: wide cost vectors go with dense reduced systems (TSP: ~sqrt(2d) rows, window ~16 d bytes)
        large_lds = _lib.MAX_LDS if d >= 4096 else 64 * 1024

        def launch_large(cap: int, band: int) -> None:
            slice_bytes = int(lib.cave_hip_large_slice_bytes(m, d, cap, band))
            if slice_bytes <= 0 or slice_bytes >= 1 << 32:
                raise HipSolverError(f"solver='hip': a cone of this size needs a {slice_bytes}-byte workspace slice "
                                     "(limit 4 GiB)")
            slots = _lib.large_slots(dev, B, slice_bytes)
            ws = _lib.workspace(dev, slots * slice_bytes)
            rc = lib.cave_hip_cone_dense_large(
                _lib.ptr(ctrs), _lib.ptr(pred), B, m, d, int(mode), float(sign), float(inner_ratio),
                int(max_iter), int(cap), large_lds, _lib.ptr(ws), slice_bytes, slots,
                _lib.ptr(out.get("proj")), _lib.ptr(out.get("rnorm")), _lib.ptr(out.get("target")),
                _lib.ptr(out.get("loss")), _lib.ptr(out.get("grad")), _lib.ptr(status), _lib.ptr(iters),
                _lib.current_stream())
            _lib.check(rc, "cave_hip_cone_dense_large")

        def run_large() -> None:
            cap, band = _large_hint.get((m, d), _large_guess(m, d))
            cap = max(cap, nnz_cap)
            for attempt in range(5):
                launch_large(cap, band)
                if not check or not bool((status == ST_TOO_LARGE).any()):
                    break
                cap, band = min(4 * cap, max(m * d, 64)), 4 * band  # dense cones: m*d non-zeros, p*p band entries
            _large_hint[(m, d)] = (cap, band)

        auto = lds_bytes == 0

        def launch_split() -> None:
            ss = _slot_store(dev, B, d)
            if ss.lds_bytes <= 0:
                raise HipSolverError("split form: no LDS configuration")
            rc = lib.cave_hip_pack_fill(_lib.ptr(ctrs), B, m, d, int(nnz_cap), 0, 4, ss.ref, 0, _lib.ptr(ss.pack_status),
                                        _lib.current_stream())
            _lib.check(rc, "cave_hip_pack_fill (slot mode)")
            rc = lib.cave_hip_cone_packed(
                ss.ref, None, _lib.ptr(pred), B, int(mode), float(sign), float(inner_ratio), int(max_iter), ss.lds_bytes, 1,
                _lib.ptr(out.get("proj")), _lib.ptr(out.get("rnorm")), _lib.ptr(out.get("target")),
                _lib.ptr(out.get("loss")), _lib.ptr(out.get("grad")), _lib.ptr(status), _lib.ptr(iters),
                _lib.current_stream())
            _lib.check(rc, "cave_hip_cone_packed (slot mode)")

        # small cones: the split form, once a checked batch of this shape has fitted it (or when this call is checked)
        if auto and waves == 0 and 0 < m and d <= SPLIT_MAX_D and B <= 2048 and (m, d) not in _tier:
            ok = _split_ok.get((m, d))
            if ok is True or (ok is None and check):
                launch_split()
                if not check:
                    return out
                fits = not bool((status == ST_TOO_LARGE).any())
                _split_ok[(m, d)] = fits
                if fits:
                    _raise_for_status(status, "solver='hip'")
                    _settled.add((m, d))
                    return out
        if auto and (m, d) not in _tier and fast_path_cannot_fit(d):
            _tier[(m, d)] = 2
        tier = _tier.get((m, d), 0) if auto else 0
        if m > 65535:
            raise HipSolverError("solver='hip': more than 65535 rows per instance are not supported")
        if tier == 2:
            run_large()
        elif tier == 1:
            cap, lds = _grow_limits(m, d)
            launch(max(cap, nnz_cap), lds, 8)  # full arena = one workgroup per CU: four waves with the wide register budget
        else:
            nw = waves if (waves != 0 or not auto) else _auto_waves(B, m, d, check)
            launch(nnz_cap, lds_bytes, nw)
            if check and auto and waves == 0 and nw == 4:
                fits = not bool((status == ST_TOO_LARGE).any())
                _wide_ok[(m, d)] = fits
                if not fits:
                    launch(nnz_cap, lds_bytes, 2)  # more than 32 reduced rows: two waves hold up to 64
        if check:
            if auto and tier == 0 and bool((status == ST_TOO_LARGE).any()):
                tier = _tier[(m, d)] = 1
                cap, lds = _grow_limits(m, d)
                launch(max(cap, nnz_cap), lds, 8)  # full arena = one workgroup per CU: four waves with the wide register budget
            if auto and tier == 1 and bool((status == ST_TOO_LARGE).any()):
                tier = _tier[(m, d)] = 2
                run_large()
            _raise_for_status(status, "solver='hip'")
            if auto:
                _settled.add((m, d))
    return out


def project_hip(tight_ctrs: torch.Tensor, signed_cost: torch.Tensor, max_iter: int = 0, nnz_cap: int = 0,
                lds_bytes: int = 0, waves: int = 0, check: bool = True) -> tuple[torch.Tensor, torch.Tensor]:
    """(proj, rnorm) on signed_cost's device and dtype — `_batch_project(..., 'nnls')` (src/cave.py:231-264).

    ``max_iter`` caps the Newton iterations of the GPU solver (0 = default 100); unlike
    Clarabel's ``max_iter`` it does not produce an interior iterate (src/cave.py:302).
    """
    o = cone_op_dense(tight_ctrs, signed_cost, MODE_PROJECT, 1.0, 0.0, max_iter=max_iter, nnz_cap=nnz_cap,
                      lds_bytes=lds_bytes, waves=waves, check=check, outputs=("proj", "rnorm"))
    device, dtype = signed_cost.device, signed_cost.dtype
    return o["proj"].to(device=device, dtype=dtype), o["rnorm"].to(device=device, dtype=dtype)


def average_ctrs_hip(tight_ctrs: torch.Tensor) -> torch.Tensor:
    """`_average_ctrs` (src/cave.py:222-228) in one streaming pass."""
    o = cone_op_dense(tight_ctrs, None, MODE_AVG, outputs=("target",))
    return o["target"].to(device=tight_ctrs.device, dtype=tight_ctrs.dtype)
