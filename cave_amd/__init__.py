"""cave_amd — MI355X (gfx950) cone-projection backend for CaVE (`solver='hip'`).

Only the hot path of khalil-research/CaVE lives here: the per-instance projection
onto the cone of tight-constraint normals and the cone-aligned cosine loss /
gradient, as hand-written HIP kernels behind a C ABI (include/cave_hip.h).
Importing the package does not touch the GPU; constructing a loss module or
calling `project_hip` does, and raises ImportError if the HIP extension or a
device is missing.
"""

from .abcmodule import EPO  # noqa: F401

__all__ = ["EPO", "exactConeAlignedCosine", "innerConeAlignedCosine", "project_hip", "average_ctrs_hip"]


def __getattr__(name):  # lazy: keep `import cave_amd` free of torch/ctypes side effects
    if name in ("exactConeAlignedCosine", "innerConeAlignedCosine", "abstractConeAlignedCosine"):
        from . import cave

        return getattr(cave, name)
    if name in ("project_hip", "average_ctrs_hip", "cone_op_dense", "HipSolverError"):
        from . import qpsolver

        return getattr(qpsolver, name)
    raise AttributeError(name)
