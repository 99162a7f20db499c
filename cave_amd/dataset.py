"""Device-resident packed cone store — the replacement for per-step dense padding.

The reference keeps each instance's tight-constraint normals as a ragged float32
matrix (`optDatasetConstrs.ctrs`, /root/reference src/dataset.py:72) and its
`collate_fn` zero-pads them to a dense (B, m_max, d) tensor every step
(src/dataset.py:133-144): 0.18 GB per step at TSP-20/B=1024, hundreds of GB for
TSP-100.  Cones are static per instance, so here they are packed ONCE on the GPU
(unit rows -> a sign byte per coordinate, +a/-a equality pairs -> one free row,
remaining rows -> CSR + CSC, plus the precomputed `_average_ctrs` vector) and a
batch is just a tensor of instance ids.

    store = ConeStore.from_ragged(dataset.ctrs)         # one-time, streams the dense form through the GPU
    ids   = torch.tensor([...])                         # what the collate_fn replacement returns
    out   = store.cone_op(ids, pred_cost, mode, sign)   # same fused kernel as the dense path
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .qpsolver import _grow_limits, _large_guess, _raise_for_status, fast_path_cannot_fit

__all__ = ["ConeStore", "PackedBatch", "collate_ids", "prefetch"]


class ConeStore:
    RB_CACHE_MAX_BYTES = 4 << 30  # cap of the large path's red-black cache (see _finalize; 128 MB for 1024 30x30 grids)

    def __init__(self, d: int, device: torch.device):
        self.d = int(d)
        self.device = device
        self.n = 0
        self.t: dict[str, torch.Tensor] = {}
        self.max_rows = 0
        self.max_nnz = 0
        self._c = None

    # ------------------------------------------------------------------ build
    @classmethod
    def from_dense(cls, tight_ctrs: torch.Tensor, chunk: int = 4096) -> "ConeStore":
        """Pack a zero-padded (N, m_max, d) tensor (host or device); streamed in chunks of instances."""
        return cls.from_chunks([tight_ctrs[i:i + chunk] for i in range(0, tight_ctrs.shape[0], chunk)])

    @classmethod
    def from_ragged(cls, ctrs: list[torch.Tensor], chunk: int = 1024) -> "ConeStore":
        """Pack `optDatasetConstrs.ctrs` (list of (m_i, d) tensors), padding one chunk at a time."""
        from torch.nn.utils.rnn import pad_sequence

        chunks = [pad_sequence(ctrs[i:i + chunk], batch_first=True, padding_value=0.0)
                  for i in range(0, len(ctrs), chunk)]
        return cls.from_chunks(chunks)

    @classmethod
    def from_ragged_shard(cls, ctrs: list[torch.Tensor], rank: int, world: int, chunk: int = 1024) -> "ConeStore":
        """This rank's shard of `optDatasetConstrs.ctrs` under data parallelism: instances
        are dealt to ranks balanced by their non-zero counts (what the packed store and the kernels' work
        scale with), not by instance count; each GPU packs and keeps only its own cones.  The same
        partition is computed on every rank.  `store.global_ids[j]` is the dataset index of store slot j;
        `store.local_ids(ids)` maps dataset indices owned by this rank to store slots."""
        from .dist import weighted_shards

        weights = [int(torch.count_nonzero(c)) for c in ctrs]
        mine = weighted_shards(weights, world)[rank]
        self = cls.from_ragged([ctrs[int(i)] for i in mine], chunk)
        self.global_ids = torch.as_tensor(mine, dtype=torch.int64)
        self.shard = (int(rank), int(world))
        return self

    def local_ids(self, dataset_ids: torch.Tensor) -> torch.Tensor:
        """Store slots of dataset indices (all of which must belong to this rank's shard)."""
        gids = getattr(self, "global_ids", None)
        ids = torch.as_tensor(dataset_ids, dtype=torch.int64).cpu()
        if gids is None:
            return ids
        pos = torch.searchsorted(gids, ids)
        if bool((pos >= gids.numel()).any()) or not torch.equal(gids[pos.clamp(max=gids.numel() - 1)], ids):
            raise IndexError("dataset index not in this rank's shard")
        return pos

    @classmethod
    def from_chunks(cls, chunks: list[torch.Tensor]) -> "ConeStore":
        return cls.from_chunks_lazy(lambda ch: ch, chunks)

    @classmethod
    def from_chunks_lazy(cls, make_chunk, keys: list) -> "ConeStore":
        """Like from_chunks, but chunk k is produced on demand by `make_chunk(keys[k])` (called twice per
        chunk: count pass and fill pass) -- for datasets whose dense form does not fit anywhere at once
        (TSP-100: 102 MB per instance)."""
        lib = _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())

        class _Chunks:
            def __iter__(self):
                return (make_chunk(k) for k in keys)

        chunks = _Chunks()
        d = None  # known once the first chunk has been produced
        stream = _lib.current_stream()
        # pass 1: counts.  Default launch limits are sized for small structured cones; a chunk that does
        # not fit is re-counted with one wave per instance and the full 160 KiB arena, then on the
        # large-cone path (global workspace).  lim = (nnz_cap, lds_bytes, waves) or ("large", nnz_cap).
        def pack_large(x, cap, n_rows, n_nnz, store, slot, status):
            B, m, _ = x.shape
            slice_bytes = int(lib.cave_hip_large_slice_bytes(m, d, cap, 1))
            slots = _lib.large_slots(dev, B, slice_bytes)
            ws = _lib.workspace(dev, slots * slice_bytes)
            _lib.check(lib.cave_hip_pack_large(_lib.ptr(x), B, m, d, cap, _lib.ptr(ws), slice_bytes, slots,
                                               _lib.ptr(n_rows), _lib.ptr(n_nnz), store, slot, _lib.ptr(status), stream),
                       "cave_hip_pack_large")

        counts, limits = [], []
        for ch in chunks:
            x = ch.to(device=dev, dtype=torch.float32).contiguous()
            del ch
            B, m, _ = x.shape
            if d is None:
                d = int(x.shape[2])
            n_rows = torch.empty(B, dtype=torch.int32, device=dev)
            n_nnz = torch.empty(B, dtype=torch.int32, device=dev)
            status = torch.empty(B, dtype=torch.int32, device=dev)
            lim = (0, 0, 0)
            tier = 2 if fast_path_cannot_fit(d) else 0
            while True:
                if tier < 2:
                    _lib.check(lib.cave_hip_pack_count(_lib.ptr(x), B, m, d, lim[0], lim[1], lim[2], _lib.ptr(n_rows),
                                                       _lib.ptr(n_nnz), _lib.ptr(status), stream), "cave_hip_pack_count")
                    if bool((status == _lib.ST_TOO_LARGE).any()):
                        tier += 1
                        if tier == 1:
                            cap, lds = _grow_limits(m, d)
                            lim = (cap, lds, 8)
                        continue
                    break
                cap = _large_guess(m, d)[0]
                for attempt in range(5):
                    pack_large(x, cap, n_rows, n_nnz, None, 0, status)
                    if not bool((status == _lib.ST_TOO_LARGE).any()):
                        break
                    cap = min(4 * cap, max(m * d, 64))
                lim = ("large", cap)
                break
            _raise_for_status(status, "ConeStore pack")
            counts.append((n_rows, n_nnz))
            limits.append(lim)
        if d is None:
            raise ValueError("ConeStore: no chunks")
        self = cls(d, dev)
        n_rows = torch.cat([c[0] for c in counts]).to(torch.int64)
        n_nnz = torch.cat([c[1] for c in counts]).to(torch.int64)
        N = int(n_rows.numel())
        z = torch.zeros(1, dtype=torch.int64, device=dev)
        row_off = torch.cat([z, torch.cumsum(n_rows, 0)])
        nnz_off = torch.cat([z, torch.cumsum(n_nnz, 0)])
        R, Z = int(row_off[-1]), int(nnz_off[-1])
        self.n = N
        self.max_rows = int(n_rows.max()) if N else 0
        self.max_nnz = int(n_nnz.max()) if N else 0
        t = self.t
        t["row_off"], t["nnz_off"] = row_off, nnz_off
        t["n_valid"] = torch.zeros(N, dtype=torch.int32, device=dev)
        t["flags"] = torch.zeros(N, dtype=torch.uint8, device=dev)
        t["usign"] = torch.zeros(N * d, dtype=torch.uint8, device=dev)
        t["avg"] = torch.zeros(N * d, dtype=torch.float32, device=dev)
        t["vkind"] = torch.zeros(max(R, 1), dtype=torch.uint8, device=dev)
        t["rlo"] = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
        t["rhi"] = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
        t["ccol"] = torch.zeros(max(Z, 1), dtype=torch.int16, device=dev)
        t["cval"] = torch.zeros(max(Z, 1), dtype=torch.float32, device=dev)
        t["cptr"] = torch.zeros(N * (d + 1), dtype=torch.int32, device=dev)
        t["cvar"] = torch.zeros(max(Z, 1), dtype=torch.int16, device=dev)
        t["cvalc"] = torch.zeros(max(Z, 1), dtype=torch.float32, device=dev)
        self._c = _lib.Store(n=N, d=d, reserved=0, **{k: v.data_ptr() for k, v in t.items()})
        # pass 2: fill
        slot = 0
        for ch, lim in zip(chunks, limits):
            x = ch.to(device=dev, dtype=torch.float32).contiguous()
            B, m, _ = x.shape
            status = torch.empty(B, dtype=torch.int32, device=dev)
            if lim[0] == "large":
                pack_large(x, lim[1], None, None, C.byref(self._c), slot, status)
            else:
                _lib.check(lib.cave_hip_pack_fill(_lib.ptr(x), B, m, d, lim[0], lim[1], lim[2], C.byref(self._c), slot,
                                                  _lib.ptr(status), stream), "cave_hip_pack_fill")
            _raise_for_status(status, "ConeStore fill")
            slot += B
        self.warm_start = False
        self.diet_min_batch = 256  # batches up to one workgroup per compute unit keep the ordinary (faster) layout
        self.fits4 = self.max_rows <= 32  # 4-wave workgroups hold reduced systems up to 32 rows
        self.waves = 0  # 0 = choose per call
        self.large_waves = 0  # large-cone path: waves per workgroup (1, 2, 4; 0 = the library's choice)
        self.all_pm1 = bool((t["flags"] & 1).all()) if N else False
        # every instance qualifies for the one-wave lite solver (cone_core.h: +-1 entries, <= 32 reduced rows,
        # <= 1536 non-zeros, d <= 256, rows ordered [free | <= 8 bound rows]; a cone with a column of more than 8 entries
        # or another row order falls back to the general solver inside the kernel)
        self.lite = self.all_pm1 and self.max_rows <= 32 and self.max_nnz <= 1536 and d <= 256
        self.lds_bytes = int(lib.cave_hip_packed_lds_bytes(d, self.max_rows, self.max_nnz, int(self.all_pm1)))
        # large batches (> 2048 instances) run the general solver: no room reserved for the lite structures
        self.lds_bytes_big = int(lib.cave_hip_packed_lds_bytes(d, self.max_rows, self.max_nnz, 2 if self.all_pm1 else 0))
        # cones beyond the LDS-resident solver (more than 64 reduced rows or too many non-zeros) run on the
        # large-cone path, which reads the store in place and keeps the Newton systems as bands
        # "diet" layout (include/cave_hip.h, cave_hip_packed_lds_bytes mode 3) for +-1 cones of the TSP-50 class: when the
        # ordinary arena allows one workgroup per compute unit only (> 80 KB) and the diet one allows two, batches
        # that fill the GPU more than once are launched with the diet figure (the kernel then takes the diet layout for
        # the instances that need it); the indices of such a store carry the signs (flags bit 1)
        self.lds_bytes_diet = 0
        if self.all_pm1 and 32 < self.max_rows <= 64 and self.lds_bytes > 80 * 1024:
            diet = int(lib.cave_hip_packed_lds_bytes(d, self.max_rows, self.max_nnz, 3))
            if 0 < diet <= 80 * 1024:
                self.lds_bytes_diet = diet
        self.large = self.lds_bytes <= 0 or self.max_rows > 64
        self.band_entries, self.max_bw = self._max_band_entries() if self.large else (0, 0)
        # LDS for the large path's hot arrays: ring window + staging buffers + three row vectors
        self.large_lds = int(lib.cave_hip_packed_large_lds_bytes(int(self.max_rows), int(self.max_bw))) if self.large else 0
        if self.large or self.lds_bytes_diet:
            self._fold_signs()
        self.rb_cache = None
        if self.large and self.all_pm1 and self.n > 0:
            # band systems of cones without bound rows (grid shortest path): what the red-black reduction derives from the
            # static cone -- independent set, recipes of the Schur complement -- is kept per instance after its first
            # projection (include/cave_hip.h rb_cache; ~140 bytes per reduced row).  Skipped beyond RB_CACHE_MAX_BYTES.
            stride = int(lib.cave_hip_packed_large_rb_bytes(int(self.max_rows)))
            if 0 < stride and self.n * stride <= self.RB_CACHE_MAX_BYTES:
                self.rb_cache = torch.zeros(self.n * stride, dtype=torch.uint8, device=dev)
                self._c.rb_cache, self._c.rb_stride = self.rb_cache.data_ptr(), stride
        self._build_lite()
        return self

    def _build_lite(self) -> None:
        """Cones are static per instance (src/dataset.py:72): when every instance qualifies for the one-wave solver,
        build its index structures ONCE -- a lite store beside the packed one (cave_hip_lite_from_packed) -- and serve
        batches of up to 2048 ids through the solve half of the step kernel (cave_hip_cone_step: 7 KB per instance in
        one memory round trip, no per-call build of the structures, the lite-only code object)."""
        from .qpsolver import _LiteSlots

        self.lite_slots = None
        if not self.lite or self.large or self.n == 0:
            return
        lib = _lib.load()
        ls = _LiteSlots(self.device, self.n, self.d)
        _lib.check(lib.cave_hip_lite_from_packed(C.byref(self._c), ls.ref, _lib.ptr(ls.pack_status), _lib.current_stream()),
                   "cave_hip_lite_from_packed")
        if bool((ls.pack_status == 0).all()):
            self.lite_slots = ls

    def _fold_signs(self) -> None:
        """Large path only (it reads the store in place, every Newton iteration): for instances whose entries are all
        +-1 (flags bit 0) put the sign into bit 15 of the 16-bit indices and set flags bit 1 -- the kernels then never
        load the fp32 value arrays of those instances (a third of the bytes and half of the loads of the streaming
        phases).  Idempotent; the value arrays stay as they are."""
        t, d = self.t, self.d
        if self.n == 0 or d >= 0x8000 or self.max_rows >= 0x8000:
            return
        pm1 = (t["flags"] & 1).bool()
        if not bool(pm1.any()):
            return
        nnz = (t["nnz_off"][1:] - t["nnz_off"][:-1]).to(torch.int64)
        per_entry = torch.repeat_interleave(pm1, nnz)  # [Z] entries of all-+-1 instances
        if per_entry.numel() != t["ccol"].numel():
            return  # (an empty store keeps one dummy entry)
        for idx, val in (("ccol", "cval"), ("cvar", "cvalc")):
            sign = ((t[val] < 0) & per_entry).to(torch.int16) << 15  # int16: bit 15 = the sign bit of the container
            t[idx] |= sign
        t["flags"] |= (pm1.to(torch.uint8) << 1)

    def _max_band_entries(self) -> int:
        """max over instances of rows * (half bandwidth + 1) of M M^T in the stored row order
        (half bandwidth = widest span of reduced-row indices meeting in one column of the CSC)."""
        t, d, N = self.t, self.d, self.n
        if N == 0 or self.max_nnz == 0:
            return 1, 0
        cptr = t["cptr"].view(N, d + 1).to(torch.int64)
        base = t["nnz_off"][:-1, None]
        lo, hi = cptr[:, :-1] + base, cptr[:, 1:] + base
        cvar = t["cvar"].to(torch.int64) & (0x7fff if self.max_rows < 0x8000 else 0xffff)  # (bit 15: see _fold_signs)
        last = cvar[(hi - 1).clamp_(min=0)]
        first = cvar[lo.clamp_(max=cvar.numel() - 1)]
        span = torch.where(hi > lo, last - first, torch.zeros_like(lo))
        bw = span.max(dim=1).values
        rows = t["row_off"][1:] - t["row_off"][:-1]
        return int((rows * (bw + 1)).max().item()), int(bw.max().item())

    # -------------------------------------------------------------------- use
    def _waves_for(self, B: int) -> int:
        """Waves per instance.  Two cooperating waves shorten the critical path while the GPU has idle
        SIMDs (about one instance per SIMD: B <= ~1024 on 256 CUs); beyond that one wave per instance
        puts more instances in flight and wins on throughput (measured crossover between 1024 and 2048)."""
        if self.waves in (1, 2, 4, 8):
            return self.waves if (self.waves != 4 or self.fits4) else 2
        if self.lite and B <= 2048:
            return 1  # small +-1 cones: the one-wave kernels carry the lite solver (cone_core.h)
        if self.lds_bytes > 80 * 1024:
            return 8  # one workgroup per CU whatever the shape: four waves with the wide register budget (64 rows)
        if self.fits4 and B <= 1280:
            return 4  # measured (TSP-20, packed): 116 vs 129 us at B = 256, 138 vs 145 us at B = 1024
        return 2 if B <= 1280 else 1

    # ------------------------------------------------------------- warm start
    def enable_warm_start(self, on: bool = True) -> None:
        """Opt-in: keep, per instance, the multipliers its last converged projection ended with (4 bytes per
        reduced row) and start the next projection of that instance there.  Cones are static per instance and
        predictions drift slowly from one epoch to the next (src/dataset.py:72), so the old active set is nearly
        right: TSP-20 training needs ~2 Newton iterations per step instead of ~5.  The projection is unique, so
        results equal those of a cold start to the solver's tolerance.  A failed solve invalidates its entry."""
        if on and "warm_theta" not in self.t:
            R = int(self.t["row_off"][-1])
            self.t["warm_theta"] = torch.zeros(max(R, 1), dtype=torch.float32, device=self.device)
            self.t["warm_state"] = torch.zeros(max(self.n, 1), dtype=torch.uint8, device=self.device)
        self.warm_start = bool(on)
        self._c.warm_theta = self.t["warm_theta"].data_ptr() if on else None
        self._c.warm_state = self.t["warm_state"].data_ptr() if on else None

    def reset_warm_start(self) -> None:
        """Forget every cached multiplier (the next projections start cold)."""
        if "warm_state" in self.t:
            self.t["warm_state"].zero_()

    def nbytes(self) -> int:
        rb = getattr(self, "rb_cache", None)
        return sum(v.numel() * v.element_size() for v in self.t.values()) + (rb.numel() if rb is not None else 0)

    def algorithmic_bytes(self, ids: torch.Tensor) -> int:
        """Bytes one projection pass must touch for these instances: packed rows + y in, proj/rnorm out
        (packed format as actually stored: 6 B per CSR entry + 6 B per CSC entry)."""
        nz = (self.t["nnz_off"][ids + 1] - self.t["nnz_off"][ids]).sum().item()
        rows = (self.t["row_off"][ids + 1] - self.t["row_off"][ids]).sum().item()
        B = ids.numel()
        return int(12 * nz + 9 * rows + B * (self.d * (1 + 4) + 4 * (self.d + 1) + 8 * self.d + 4))

    def cone_op(self, ids: torch.Tensor, pred_cost: torch.Tensor | None, mode: int, sign: float = 1.0,
                inner_ratio: float = 0.2, *, max_iter: int = 0, check: bool = True, zero_failed: bool = False,
                outputs: tuple[str, ...] = ("proj", "rnorm")) -> dict[str, torch.Tensor]:
        """`zero_failed`: ask the kernel to write loss 0 / gradient 0 for instances whose status is not OK (the lite
        slots' kernel can: out["zero_failed"] is then set; callers mask the others themselves)."""
        lib = _lib.load()
        dev = self.device
        # (every conversion below is skipped when the tensor already is what the kernel reads: a training step is
        #  host-bound, and each no-op .to() / .contiguous() costs a microsecond or two of it)
        if ids.device != dev or ids.dtype != torch.int64 or not ids.is_contiguous():
            ids = ids.to(device=dev, dtype=torch.int64).contiguous()
        B, d = int(ids.numel()), self.d
        pred = None
        if pred_cost is not None:
            pred = pred_cost.detach()
            if pred.device != dev or pred.dtype != torch.float32 or not pred.is_contiguous():
                pred = pred.to(device=dev, dtype=torch.float32).contiguous()
        out: dict[str, torch.Tensor] = {}
        other_device = torch.cuda.current_device() != dev.index
        if other_device:
            ctx = torch.cuda.device(dev)
            ctx.__enter__()
        try:
            for name in outputs:
                out[name] = torch.empty((B,) if name in ("rnorm", "loss") else (B, d), dtype=torch.float32, device=dev)
            status = torch.empty(B, dtype=torch.int32, device=dev)
            iters = torch.empty(B, dtype=torch.int32, device=dev)
            out["status"], out["iters"] = status, iters
            if B == 0:
                return out
            if self.large:
                slice_bytes = int(lib.cave_hip_packed_large_slice_bytes(d, self.max_rows, self.band_entries))
                slots = _lib.large_slots(dev, B, slice_bytes)
                ws = _lib.workspace(dev, slots * slice_bytes)
                rc = lib.cave_hip_cone_packed_large(
                    C.byref(self._c), _lib.ptr(ids), _lib.ptr(pred), B, int(mode), float(sign), float(inner_ratio),
                    int(max_iter), self.large_lds, int(self.large_waves), _lib.ptr(ws), slice_bytes, slots,
                    _lib.ptr(out.get("proj")), _lib.ptr(out.get("rnorm")), _lib.ptr(out.get("target")),
                    _lib.ptr(out.get("loss")), _lib.ptr(out.get("grad")), _lib.ptr(status), _lib.ptr(iters),
                    _lib.current_stream())
                _lib.check(rc, "cave_hip_cone_packed_large")
            elif (self.lite_slots is not None and B <= 2048 and mode != _lib.MODE_INNER_IPM and not self.warm_start
                  and self.waves == 0):
                from .qpsolver import _launch_step

                _launch_step(self.lite_slots, pred, B, mode, sign, inner_ratio, max_iter, out, status, iters, None, None, ids=ids,
                             zero_failed=zero_failed)
                if zero_failed:
                    out["zero_failed"] = True
            else:
                lds = self.lds_bytes if B <= 2048 else self.lds_bytes_big
                if self.lds_bytes_diet and B > self.diet_min_batch and mode != _lib.MODE_INNER_IPM and self.waves in (0, 8):
                    lds = self.lds_bytes_diet  # two workgroups per compute unit
                rc = lib.cave_hip_cone_packed(
                    C.byref(self._c), _lib.ptr(ids), _lib.ptr(pred), B, int(mode), float(sign), float(inner_ratio),
                    int(max_iter), lds, self._waves_for(B),
                    _lib.ptr(out.get("proj")), _lib.ptr(out.get("rnorm")), _lib.ptr(out.get("target")),
                    _lib.ptr(out.get("loss")), _lib.ptr(out.get("grad")), _lib.ptr(status), _lib.ptr(iters),
                    _lib.current_stream())
                _lib.check(rc, "cave_hip_cone_packed")
            self.last_iters = iters  # Newton iterations / status of the most recent call (device tensors; diagnostics)
            self.last_status = status
            if check:
                _raise_for_status(status, "solver='hip' (packed)")
        finally:
            if other_device:
                ctx.__exit__(None, None, None)
        return out


class PackedBatch:
    """What a loss module receives instead of the dense (B, m_max, d) tensor: store + instance ids."""

    def __init__(self, store: ConeStore, ids: torch.Tensor):
        self.store = store
        self.ids = ids


def collate_ids(batch):
    """Drop-in for the reference `collate_fn` (src/dataset.py:133-144) when the dataset yields
    (x, c, w, z, instance_id): stacks the dense fields and returns the ids instead of padded cones."""
    x, c, w, z, ids = zip(*batch)
    return (torch.stack(x, 0), torch.stack(c, 0), torch.stack(w, 0), torch.stack(z, 0),
            torch.as_tensor(ids, dtype=torch.int64))


def prefetch(loader, slot: int = -1, device=None):
    """Wrap a DataLoader whose batches carry the dense `tight_ctrs` in field `slot` (the reference's collate_fn puts
    it last: src/dataset.py:133-144) so that the loop body of code_sample.py:48-60 stays as it is,

        for data in prefetch(loader):
            x, c, w, z, bctr = data
            x, c, w, z, bctr = x.cuda(), c.cuda(), w.cuda(), z.cuda(), bctr.cuda()
            loss = cave(reg(x), bctr); ...

    and gets the fused step: `bctr` is a PreparedCones whose reduced cones are already on the device, and the loss call
    of batch i packs batch i+1 in the same launch (qpsolver.cone_op_prepared).  One batch of look-ahead: the generator
    pulls batch i+1 from the loader before it yields batch i.  Cones are moved to `device` (default: the current HIP
    device) here; shapes the fused form does not take pass through as tensors."""
    from .qpsolver import PreparedCones, prepare_dense

    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def cones_of(batch):
        return batch[slot].to(dev, non_blocking=True)

    it = iter(loader)
    try:
        cur = next(it)
    except StopIteration:
        return
    cur_cones = prepare_dense(cones_of(cur))
    while True:
        try:
            nxt = next(it)
        except StopIteration:
            nxt = None
        nxt_dense = cones_of(nxt) if nxt is not None else None
        if isinstance(cur_cones, PreparedCones) and nxt_dense is not None:
            cur_cones.then(nxt_dense)
        out = list(cur)
        out[slot] = cur_cones
        yield type(cur)(out) if isinstance(cur, (tuple, list)) else out
        if nxt is None:
            return
        if isinstance(cur_cones, PreparedCones) and cur_cones.next is not None:
            nxt_cones = cur_cones.next          # packed by the loss call of the batch just yielded
        else:                                   # (no loss call happened on it, or the shape does not qualify)
            nxt_cones = prepare_dense(nxt_dense)
        cur, cur_cones = nxt, nxt_cones
