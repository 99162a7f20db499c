"""ctypes binding of the C ABI in include/cave_hip.h (libcave_hip.so, gfx950).

PyTorch is plumbing here: it owns device memory and the HIP stream; every call
passes raw `data_ptr()`s and `torch.cuda.current_stream().cuda_stream` across
the C boundary.  There is NO CPU fallback: if the shared library is missing or
no HIP device is visible, `load()` raises ImportError (mirroring how the
reference refuses solver='clarabel' without cvxpy, src/cave.py:113-117).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libcave_hip.so")
_CSRC = os.path.join(_HERE, "csrc")
_HEADERS = [os.path.join(_CSRC, n) for n in ("kernels.h", "cone_common.h", "cone_core.h", "cone_band.h", "cone_dense.h", "cone_rb.h", "cone_instance.h", "cone_step.h",
                                             "wave_prims.h", "ctx_wave.h", "ctx_block.h")] + \
    [os.path.join(_ROOT, "include", "cave_hip.h")]
# translation units: the C ABI (host code) + one file per kernel shape (cave_amd/csrc/kernels.h)
_UNITS = ["cave_hip"] + [f"k_{op}_w{w}" for op in ("dense", "pack", "packed") for w in (1, 2, 4, 8)] + \
    ["k_large_dense", "k_large_pack", "k_large_packed_w1", "k_large_packed_w2", "k_large_packed_w4", "k_step"]
_SOURCES = [os.path.join(_CSRC, u + ".hip") for u in _UNITS] + _HEADERS
_OBJ_DIR = os.path.join(_CSRC, "build")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]

# status / mode constants (include/cave_hip.h)
ST_OK, ST_NOT_CONVERGED, ST_TOO_LARGE, ST_BAD_INPUT = 0, 1, 2, 3
MODE_PROJECT, MODE_EXACT, MODE_INNER, MODE_HEURISTIC, MODE_AVG, MODE_INNER_IPM = 0, 1, 2, 3, 4, 5
MAX_LDS = 160 * 1024

ABI_SYMBOLS = (
    "cave_hip_version", "cave_hip_last_error", "cave_hip_device_count", "cave_hip_default_limits",
    "cave_hip_cone_dense", "cave_hip_pack_count", "cave_hip_pack_fill", "cave_hip_cone_packed",
    "cave_hip_packed_lds_bytes",
    "cave_hip_large_slice_bytes", "cave_hip_packed_large_slice_bytes", "cave_hip_cone_dense_large",
    "cave_hip_pack_large", "cave_hip_cone_packed_large", "cave_hip_packed_large_lds_bytes", "cave_hip_packed_large_rb_bytes",
    "cave_hip_step_lds_bytes", "cave_hip_cone_step", "cave_hip_lite_from_packed",
)


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """hipcc --offload-arch=gfx950 -> cave_amd/libcave_hip.so (in-tree; cross-compiles without a GPU).

    Every kernel shape is its own translation unit (objects under cave_amd/csrc/build/): they compile in
    parallel and only the units older than their sources are rebuilt."""
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(_OBJ_DIR, exist_ok=True)
    newest_hdr = max(os.path.getmtime(p) for p in _HEADERS)
    todo, objs = [], []
    for u in _UNITS:
        src, obj = os.path.join(_CSRC, u + ".hip"), os.path.join(_OBJ_DIR, u + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_hdr, os.path.getmtime(src)):
            todo.append((src, obj))
    if not todo and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *HIPCC_FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs is None:
        jobs = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4, 8))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


class Store(C.Structure):
    """struct cave_cone_store (include/cave_hip.h)."""
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


class LiteStore(C.Structure):
    """struct cave_lite_store (include/cave_hip.h): transient per-batch store of the fused step kernel."""
    _fields_ = [
        ("n", C.c_int64), ("d", C.c_int32), ("reserved", C.c_int32),
        ("hdr", C.c_void_p), ("usign", C.c_void_p), ("avg", C.c_void_p), ("rowptr", C.c_void_p),
        ("ell", C.c_void_p), ("csr16", C.c_void_p), ("rl", C.c_void_p),
    ]


_lib = None


def load_library() -> C.CDLL:
    """dlopen the shared library and declare signatures.  Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "solver='hip' has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    lib.cave_hip_version.restype = C.c_int32
    lib.cave_hip_last_error.restype = C.c_char_p
    lib.cave_hip_device_count.restype = C.c_int32
    for name in ABI_SYMBOLS:
        getattr(lib, name)  # AttributeError here = ABI drift between header and library
    i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
    lib.cave_hip_default_limits.argtypes = [i64, i64, C.POINTER(i32), C.POINTER(i32)]
    lib.cave_hip_cone_dense.argtypes = [vp, vp, i64, i64, i64, i32, f32, f32, i32, i32, i32, i32,
                                        vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cave_hip_pack_count.argtypes = [vp, i64, i64, i64, i32, i32, i32, vp, vp, vp, vp]
    lib.cave_hip_pack_fill.argtypes = [vp, i64, i64, i64, i32, i32, i32, C.POINTER(Store), i64, vp, vp]
    lib.cave_hip_cone_packed.argtypes = [C.POINTER(Store), vp, vp, i64, i32, f32, f32, i32, i32, i32,
                                         vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cave_hip_packed_lds_bytes.argtypes = [i64, i32, i32, i32]
    for name in ("cave_hip_default_limits", "cave_hip_cone_dense", "cave_hip_pack_count", "cave_hip_pack_fill",
                 "cave_hip_cone_packed", "cave_hip_packed_lds_bytes"):
        getattr(lib, name).restype = C.c_int32
    lib.cave_hip_large_slice_bytes.argtypes = [i64, i64, i64, i64]
    lib.cave_hip_large_slice_bytes.restype = i64
    lib.cave_hip_packed_large_slice_bytes.argtypes = [i64, i64, i64]
    lib.cave_hip_packed_large_slice_bytes.restype = i64
    lib.cave_hip_packed_large_lds_bytes.argtypes = [i32, i32]
    lib.cave_hip_packed_large_lds_bytes.restype = i32
    lib.cave_hip_packed_large_rb_bytes.argtypes = [i64]
    lib.cave_hip_packed_large_rb_bytes.restype = i64
    lib.cave_hip_cone_dense_large.argtypes = [vp, vp, i64, i64, i64, i32, f32, f32, i32, i64, i32, vp, i64, i32,
                                              vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cave_hip_pack_large.argtypes = [vp, i64, i64, i64, i64, vp, i64, i32, vp, vp, C.POINTER(Store), i64, vp, vp]
    lib.cave_hip_cone_packed_large.argtypes = [C.POINTER(Store), vp, vp, i64, i32, f32, f32, i32, i32, i32, vp, i64, i32,
                                               vp, vp, vp, vp, vp, vp, vp, vp]
    for name in ("cave_hip_cone_dense_large", "cave_hip_pack_large", "cave_hip_cone_packed_large"):
        getattr(lib, name).restype = C.c_int32
    lib.cave_hip_step_lds_bytes.argtypes = [i64, i64]
    lib.cave_hip_step_lds_bytes.restype = i32
    lib.cave_hip_cone_step.argtypes = [C.POINTER(LiteStore), vp, vp, i64, i32, f32, f32, i32, i32, vp, vp, vp, vp, vp, vp, vp,
                                       vp, i64, i64, i64, C.POINTER(LiteStore), vp, vp, vp]
    lib.cave_hip_cone_step.restype = i32
    lib.cave_hip_lite_from_packed.argtypes = [C.POINTER(Store), C.POINTER(LiteStore), vp, vp]
    lib.cave_hip_lite_from_packed.restype = i32
    _lib = lib
    return lib


def load() -> C.CDLL:
    """Library + a visible HIP device, or ImportError."""
    lib = load_library()
    import torch

    if not torch.cuda.is_available() or lib.cave_hip_device_count() <= 0:
        raise ImportError("solver='hip' needs a visible HIP device (MI355X / gfx950); none found. "
                          "There is no CPU fallback in cave_amd.")
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().cave_hip_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def default_limits(m_max: int, d: int) -> tuple[int, int]:
    lib = load_library()
    cap, lds = C.c_int32(0), C.c_int32(0)
    check(lib.cave_hip_default_limits(m_max, d, C.byref(cap), C.byref(lds)), "cave_hip_default_limits")
    return int(cap.value), int(lds.value)


_workspaces: dict = {}


def workspace(device, nbytes: int):
    """Grow-only per-device scratch buffer for the large-cone path (caller-owned workspace of the C ABI)."""
    import torch

    t = _workspaces.get(device)
    if t is None or t.numel() < nbytes:
        _workspaces.pop(device, None)
        t = None
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[device] = t
    return t


def large_slots(device, B: int, slice_bytes: int) -> int:
    """Workgroups (= workspace slices) for a large-cone launch: 4 per CU, bounded by free memory.
    The free-memory query is a blocking driver call (~5 ms: more than a TSP-100 batch takes), so it is only made when
    the workspace HAS TO GROW (a rare path) -- and then always: a figure remembered from an earlier, emptier device
    would let a later, larger request run into an out-of-memory error."""
    import torch

    want = int(max(1, min(B, 1024)))
    held = _workspaces[device].numel() if device in _workspaces else 0
    if want * slice_bytes <= held:
        return want
    free, _ = torch.cuda.mem_get_info(device)
    budget = min((free + held) // 2, 64 << 30)
    cap = int(max(1, min(1024, budget // max(slice_bytes, 1))))
    return int(max(1, min(want, cap)))


def ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def current_stream() -> C.c_void_p:
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
