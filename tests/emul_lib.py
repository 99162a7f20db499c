"""ctypes loader for the serial gcc build of cave_amd/csrc (TEST INFRASTRUCTURE ONLY).

The build in tests/emul runs the per-instance algorithm code shared with the
HIP kernels on one serial lane, on host memory.  It exists so the CPU-only test
tier (and ASan/UBSan) can exercise that code; the cave_amd package never
imports it.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "emul", "emul_abi.cpp")
_DEPS = [
    _SRC,
    os.path.join(_HERE, "..", "cave_amd", "csrc", "cone_band.h"),
    os.path.join(_HERE, "..", "cave_amd", "csrc", "cone_dense.h"),
    os.path.join(_HERE, "..", "cave_amd", "csrc", "cone_rb.h"),
    os.path.join(_HERE, "emul", "ctx_serial.h"),
    os.path.join(_HERE, "..", "cave_amd", "csrc", "cone_core.h"),
    os.path.join(_HERE, "..", "cave_amd", "csrc", "cone_common.h"),
    os.path.join(_HERE, "..", "cave_amd", "csrc", "cone_instance.h"),
    os.path.join(_HERE, "..", "include", "cave_hip.h"),
]


def build(asan: bool = False) -> str:
    out = os.path.join(_HERE, "emul", "_emul_asan.so" if asan else "_emul.so")
    newest = max(os.path.getmtime(p) for p in _DEPS)
    if os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if asan else ["-O2"]
    cmd = ["g++", "-std=c++17", "-fPIC", "-shared", "-w", *flags, _SRC, "-o", out]
    subprocess.run(cmd, check=True)
    return out


class Store(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("d", C.c_int32), ("reserved", C.c_int32),
        ("row_off", C.c_void_p), ("nnz_off", C.c_void_p), ("n_valid", C.c_void_p), ("flags", C.c_void_p),
        ("usign", C.c_void_p), ("avg", C.c_void_p), ("vkind", C.c_void_p),
        ("rlo", C.c_void_p), ("rhi", C.c_void_p), ("ccol", C.c_void_p), ("cval", C.c_void_p),
        ("cptr", C.c_void_p), ("cvar", C.c_void_p), ("cvalc", C.c_void_p),
        ("n_rows", C.c_void_p), ("n_nnz", C.c_void_p),  # slot mode only (NULL in an exact-fit store)
        ("warm_theta", C.c_void_p), ("warm_state", C.c_void_p),  # warm start (NULL: off)
        ("rb_cache", C.c_void_p), ("rb_stride", C.c_int64),  # red-black cache of the large path (NULL: off)
    ]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Emul:
    def __init__(self, asan: bool = False):
        self.lib = C.CDLL(build(asan))

    def path_counters(self):
        """[dense-LDL^T instances, ...] since the last call (tests assert that a case took the path it is meant for)."""
        out = (C.c_long * 8)()
        self.lib.cave_emul_path_counters(out)
        return list(out)

    def cone_dense(self, ctrs, pred, mode, sign=-1.0, inner_ratio=0.2, max_iter=0, nnz_cap=0, lds_bytes=0):
        ctrs = np.ascontiguousarray(ctrs, dtype=np.float32)
        B, m, d = ctrs.shape
        pred = None if pred is None else np.ascontiguousarray(pred, dtype=np.float32)
        out = {
            "proj": np.zeros((B, d), np.float32), "rnorm": np.zeros(B, np.float32),
            "target": np.zeros((B, d), np.float32), "loss": np.zeros(B, np.float32),
            "grad": np.zeros((B, d), np.float32), "status": np.zeros(B, np.int32),
            "iters": np.zeros(B, np.int32),
        }
        rc = self.lib.cave_emul_cone_dense(
            _p(ctrs), _p(pred), C.c_int64(B), C.c_int64(m), C.c_int64(d), C.c_int32(mode),
            C.c_float(sign), C.c_float(inner_ratio), C.c_int32(max_iter), C.c_int32(nnz_cap), C.c_int32(lds_bytes),
            _p(out["proj"]), _p(out["rnorm"]), _p(out["target"]), _p(out["loss"]), _p(out["grad"]),
            _p(out["status"]), _p(out["iters"]))
        assert rc == 0, rc
        return out

    def pack(self, ctrs, nnz_cap=0, lds_bytes=0):
        ctrs = np.ascontiguousarray(ctrs, dtype=np.float32)
        B, m, d = ctrs.shape
        n_rows = np.zeros(B, np.int32); n_nnz = np.zeros(B, np.int32); status = np.zeros(B, np.int32)
        rc = self.lib.cave_emul_pack_count(_p(ctrs), C.c_int64(B), C.c_int64(m), C.c_int64(d), C.c_int32(nnz_cap),
                                           C.c_int32(lds_bytes), _p(n_rows), _p(n_nnz), _p(status))
        assert rc == 0 and (status == 0).all(), (rc, status)
        row_off = np.concatenate([[0], np.cumsum(n_rows, dtype=np.int64)]).astype(np.int64)
        nnz_off = np.concatenate([[0], np.cumsum(n_nnz, dtype=np.int64)]).astype(np.int64)
        R, Z = int(row_off[-1]), int(nnz_off[-1])
        arrs = {
            "row_off": row_off, "nnz_off": nnz_off, "n_valid": np.zeros(B, np.int32), "flags": np.zeros(B, np.uint8),
            "usign": np.zeros(B * d, np.uint8), "avg": np.zeros(B * d, np.float32),
            "vkind": np.zeros(max(R, 1), np.uint8), "rlo": np.zeros(max(R, 1), np.uint32),
            "rhi": np.zeros(max(R, 1), np.uint32), "ccol": np.zeros(max(Z, 1), np.uint16),
            "cval": np.zeros(max(Z, 1), np.float32), "cptr": np.zeros(B * (d + 1), np.uint32),
            "cvar": np.zeros(max(Z, 1), np.uint16), "cvalc": np.zeros(max(Z, 1), np.float32),
        }
        st = Store(n=B, d=d, reserved=0, **{k: v.ctypes.data for k, v in arrs.items()})
        rc = self.lib.cave_emul_pack_fill(_p(ctrs), C.c_int64(B), C.c_int64(m), C.c_int64(d), C.c_int32(nnz_cap),
                                          C.c_int32(lds_bytes), C.byref(st), C.c_int64(0), _p(status))
        assert rc == 0 and (status == 0).all(), (rc, status)
        return st, arrs, int(n_rows.max(initial=0)), int(n_nnz.max(initial=0))

    def cone_packed(self, store, arrs, max_rows, max_nnz, ids, pred, mode, sign=-1.0, inner_ratio=0.2, max_iter=0):
        d = store.d
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        B = len(ids)
        pred = None if pred is None else np.ascontiguousarray(pred, dtype=np.float32)
        all_pm1 = int(bool((arrs["flags"] & 1).all()))
        lds = self.lib.cave_emul_packed_lds_bytes(C.c_int64(d), C.c_int32(max_rows), C.c_int32(max_nnz), C.c_int32(all_pm1))
        assert lds > 0
        out = {
            "proj": np.zeros((B, d), np.float32), "rnorm": np.zeros(B, np.float32),
            "target": np.zeros((B, d), np.float32), "loss": np.zeros(B, np.float32),
            "grad": np.zeros((B, d), np.float32), "status": np.zeros(B, np.int32),
            "iters": np.zeros(B, np.int32),
        }
        rc = self.lib.cave_emul_cone_packed(
            C.byref(store), _p(ids), _p(pred), C.c_int64(B), C.c_int32(mode), C.c_float(sign),
            C.c_float(inner_ratio), C.c_int32(max_iter), C.c_int32(lds),
            _p(out["proj"]), _p(out["rnorm"]), _p(out["target"]), _p(out["loss"]), _p(out["grad"]),
            _p(out["status"]), _p(out["iters"]))
        assert rc == 0, rc
        return out

    # ---- large-cone path (global-workspace arena, band Newton systems)
    def _outs(self, B, d):
        return {
            "proj": np.zeros((B, d), np.float32), "rnorm": np.zeros(B, np.float32),
            "target": np.zeros((B, d), np.float32), "loss": np.zeros(B, np.float32),
            "grad": np.zeros((B, d), np.float32), "status": np.zeros(B, np.int32),
            "iters": np.zeros(B, np.int32),
        }

    def large_slice_bytes(self, m, d, nnz_cap, band):
        self.lib.cave_emul_large_slice_bytes.restype = C.c_int64
        return int(self.lib.cave_emul_large_slice_bytes(C.c_int64(m), C.c_int64(d), C.c_int64(nnz_cap), C.c_int64(band)))

    def cone_dense_large(self, ctrs, pred, mode, sign=-1.0, inner_ratio=0.2, max_iter=0, nnz_cap=0, band=0,
                         lds_bytes=64 * 1024, slice_bytes=0):
        ctrs = np.ascontiguousarray(ctrs, dtype=np.float32)
        B, m, d = ctrs.shape
        pred = None if pred is None else np.ascontiguousarray(pred, dtype=np.float32)
        nnz_cap = nnz_cap or max(64, int((ctrs != 0).reshape(B, -1).sum(1).max(initial=0)))
        band = band or 32 * d + 4096
        slice_bytes = slice_bytes or self.large_slice_bytes(m, d, nnz_cap, band)
        out = self._outs(B, d)
        rc = self.lib.cave_emul_cone_dense_large(
            _p(ctrs), _p(pred), C.c_int64(B), C.c_int64(m), C.c_int64(d), C.c_int32(mode), C.c_float(sign),
            C.c_float(inner_ratio), C.c_int32(max_iter), C.c_int64(nnz_cap), C.c_int32(lds_bytes), C.c_int64(slice_bytes),
            _p(out["proj"]), _p(out["rnorm"]), _p(out["target"]), _p(out["loss"]), _p(out["grad"]),
            _p(out["status"]), _p(out["iters"]))
        assert rc == 0, rc
        return out

    def pack_large(self, ctrs, nnz_cap=0, band=0):
        ctrs = np.ascontiguousarray(ctrs, dtype=np.float32)
        B, m, d = ctrs.shape
        nnz_cap = nnz_cap or max(64, int((ctrs != 0).reshape(B, -1).sum(1).max(initial=0)))
        slice_bytes = self.large_slice_bytes(m, d, nnz_cap, band or 1)
        n_rows = np.zeros(B, np.int32); n_nnz = np.zeros(B, np.int32); status = np.zeros(B, np.int32)
        rc = self.lib.cave_emul_pack_large(_p(ctrs), C.c_int64(B), C.c_int64(m), C.c_int64(d), C.c_int64(nnz_cap),
                                           C.c_int64(slice_bytes), _p(n_rows), _p(n_nnz), None, C.c_int64(0), _p(status))
        assert rc == 0 and (status == 0).all(), (rc, status)
        row_off = np.concatenate([[0], np.cumsum(n_rows, dtype=np.int64)]).astype(np.int64)
        nnz_off = np.concatenate([[0], np.cumsum(n_nnz, dtype=np.int64)]).astype(np.int64)
        R, Z = int(row_off[-1]), int(nnz_off[-1])
        arrs = {
            "row_off": row_off, "nnz_off": nnz_off, "n_valid": np.zeros(B, np.int32), "flags": np.zeros(B, np.uint8),
            "usign": np.zeros(B * d, np.uint8), "avg": np.zeros(B * d, np.float32),
            "vkind": np.zeros(max(R, 1), np.uint8), "rlo": np.zeros(max(R, 1), np.uint32),
            "rhi": np.zeros(max(R, 1), np.uint32), "ccol": np.zeros(max(Z, 1), np.uint16),
            "cval": np.zeros(max(Z, 1), np.float32), "cptr": np.zeros(B * (d + 1), np.uint32),
            "cvar": np.zeros(max(Z, 1), np.uint16), "cvalc": np.zeros(max(Z, 1), np.float32),
        }
        st = Store(n=B, d=d, reserved=0, **{k: v.ctypes.data for k, v in arrs.items()})
        rc = self.lib.cave_emul_pack_large(_p(ctrs), C.c_int64(B), C.c_int64(m), C.c_int64(d), C.c_int64(nnz_cap),
                                           C.c_int64(slice_bytes), None, None, C.byref(st), C.c_int64(0), _p(status))
        assert rc == 0 and (status == 0).all(), (rc, status)
        return st, arrs, int(n_rows.max(initial=0)), int(n_nnz.max(initial=0))

    def cone_packed_large(self, store, arrs, max_rows, ids, pred, mode, sign=-1.0, inner_ratio=0.2, max_iter=0,
                          band=0, lds_bytes=64 * 1024):
        d = store.d
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        B = len(ids)
        pred = None if pred is None else np.ascontiguousarray(pred, dtype=np.float32)
        self.lib.cave_emul_packed_large_slice_bytes.restype = C.c_int64
        band = band or max_rows * max_rows
        slice_bytes = int(self.lib.cave_emul_packed_large_slice_bytes(C.c_int64(d), C.c_int64(max_rows), C.c_int64(band)))
        out = self._outs(B, d)
        rc = self.lib.cave_emul_cone_packed_large(
            C.byref(store), _p(ids), _p(pred), C.c_int64(B), C.c_int32(mode), C.c_float(sign), C.c_float(inner_ratio),
            C.c_int32(max_iter), C.c_int32(lds_bytes), C.c_int64(slice_bytes),
            _p(out["proj"]), _p(out["rnorm"]), _p(out["target"]), _p(out["loss"]), _p(out["grad"]),
            _p(out["status"]), _p(out["iters"]))
        assert rc == 0, rc
        return out


# ---------------------------------------------------------------------------------------------------------------
# SIMT emulation build (tests/emul/simt_abi.cpp): the GPU-only code paths -- wave contexts, DPP / readlane /
# ballot primitives, the one-wave lite solver, the one-/two-wave band elimination, the blocked dense LDL^T --
# compiled by g++ against a shim <hip/hip_runtime.h> in which every lane is a fiber.  TEST INFRASTRUCTURE ONLY.
_SIMT_SRC = os.path.join(_HERE, "emul", "simt_abi.cpp")
_SIMT_DEPS = _DEPS + [_SIMT_SRC, os.path.join(_HERE, "emul", "simt", "hip", "hip_runtime.h")] + [
    os.path.join(_HERE, "..", "cave_amd", "csrc", n) for n in ("wave_prims.h", "ctx_wave.h", "ctx_block.h")]


def build_simt(asan: bool = False, defines: tuple = (), tag: str = "") -> str:
    """`defines` / `tag`: a variant build (e.g. a tiny spin limit + a withheld hand-over flag) under its own name."""
    out = os.path.join(_HERE, "emul", f"_simt{tag}_asan.so" if asan else f"_simt{tag}.so")
    newest = max(os.path.getmtime(p) for p in _SIMT_DEPS)
    if os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if asan else ["-O2"]
    cmd = ["g++", "-std=c++17", "-fPIC", "-shared", "-w", *flags, *["-D" + x for x in defines],
           "-I" + os.path.join(_HERE, "emul", "simt"), _SIMT_SRC, "-o", out]
    subprocess.run(cmd, check=True)
    return out


class Simt:
    """Runs the kernels' per-instance code with their real workgroup shapes (1 / 2 / 4 waves of 64 lanes).
    Stores come from Emul.pack / Emul.pack_large (host arrays).  `seed` != 0 shuffles the order in which the lanes
    run between two rendezvous (a hand-over the source does not order then shows up as wrong numbers)."""

    def __init__(self, asan: bool = False, defines: tuple = (), tag: str = ""):
        self.lib = C.CDLL(build_simt(asan, defines, tag))
        self.lib.cave_simt_packed_large_slice_bytes.restype = C.c_int64

    def path_counters(self):
        out = (C.c_long * 8)()
        self.lib.cave_simt_path_counters(out)
        return list(out)

    def _outs(self, B, d):
        return {
            "proj": np.zeros((B, d), np.float32), "rnorm": np.zeros(B, np.float32),
            "target": np.zeros((B, d), np.float32), "loss": np.zeros(B, np.float32),
            "grad": np.zeros((B, d), np.float32), "status": np.zeros(B, np.int32),
            "iters": np.zeros(B, np.int32),
        }

    def cone_dense(self, ctrs, pred, mode, sign=-1.0, inner_ratio=0.2, max_iter=0, waves=1, seed=0):
        ctrs = np.ascontiguousarray(ctrs, dtype=np.float32)
        B, m, d = ctrs.shape
        pred = None if pred is None else np.ascontiguousarray(pred, dtype=np.float32)
        out = self._outs(B, d)
        rc = self.lib.cave_simt_cone_dense(
            _p(ctrs), _p(pred), C.c_int64(B), C.c_int64(m), C.c_int64(d), C.c_int32(mode), C.c_float(sign),
            C.c_float(inner_ratio), C.c_int32(max_iter), C.c_int32(0), C.c_int32(0), C.c_int32(waves), C.c_uint64(seed),
            _p(out["proj"]), _p(out["rnorm"]), _p(out["target"]), _p(out["loss"]), _p(out["grad"]),
            _p(out["status"]), _p(out["iters"]))
        assert rc == 0, rc
        return out

    def cone_packed(self, store, arrs, max_rows, max_nnz, ids, pred, mode, sign=-1.0, inner_ratio=0.2, max_iter=0,
                    waves=1, seed=0, diet=False):
        """diet: launch with the LDS figure of the diet layout (cave_hip_packed_lds_bytes mode 3): instances whose
        ordinary arena exceeds it take that layout (waves = 8 only; the store must carry the signs in its indices)."""
        d = store.d
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        B = len(ids)
        pred = None if pred is None else np.ascontiguousarray(pred, dtype=np.float32)
        all_pm1 = int(bool((arrs["flags"] & 1).all()))
        lds = self.lib.cave_simt_packed_lds_bytes(C.c_int64(d), C.c_int32(max_rows), C.c_int32(max_nnz),
                                                  C.c_int32(3 if diet else all_pm1))
        assert lds > 0
        out = self._outs(B, d)
        rc = self.lib.cave_simt_cone_packed(
            C.byref(store), _p(ids), _p(pred), C.c_int64(B), C.c_int32(mode), C.c_float(sign), C.c_float(inner_ratio),
            C.c_int32(max_iter), C.c_int32(lds), C.c_int32(waves), C.c_uint64(seed),
            _p(out["proj"]), _p(out["rnorm"]), _p(out["target"]), _p(out["loss"]), _p(out["grad"]),
            _p(out["status"]), _p(out["iters"]))
        assert rc == 0, rc
        return out

    def cone_packed_large(self, store, arrs, max_rows, max_bw, ids, pred, mode, sign=-1.0, inner_ratio=0.2, max_iter=0,
                          waves=2, seed=0, lds_bytes=0):
        d = store.d
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        B = len(ids)
        pred = None if pred is None else np.ascontiguousarray(pred, dtype=np.float32)
        band = max_rows * (max_bw + 1)
        slice_bytes = int(self.lib.cave_simt_packed_large_slice_bytes(C.c_int64(d), C.c_int64(max_rows), C.c_int64(band)))
        lds = lds_bytes or int(self.lib.cave_simt_packed_large_lds_bytes(C.c_int32(max_rows), C.c_int32(max_bw)))
        out = self._outs(B, d)
        rc = self.lib.cave_simt_cone_packed_large(
            C.byref(store), _p(ids), _p(pred), C.c_int64(B), C.c_int32(mode), C.c_float(sign), C.c_float(inner_ratio),
            C.c_int32(max_iter), C.c_int32(lds), C.c_int32(waves), C.c_int64(slice_bytes), C.c_uint64(seed),
            _p(out["proj"]), _p(out["rnorm"]), _p(out["target"]), _p(out["loss"]), _p(out["grad"]),
            _p(out["status"]), _p(out["iters"]))
        assert rc == 0, rc
        return out


def store_bandwidth(arrs, B, d):
    """max over instances of the half bandwidth of M M^T in the stored row order (as ConeStore._max_band_entries)."""
    bw = 0
    for b in range(B):
        cp = arrs["cptr"][b * (d + 1):(b + 1) * (d + 1)].astype(np.int64) + int(arrs["nnz_off"][b])
        for k in range(d):
            if cp[k + 1] > cp[k] + 1:
                bw = max(bw, int(arrs["cvar"][cp[k + 1] - 1]) - int(arrs["cvar"][cp[k]]))
    return bw
