"""Base-class contract the CaVE loss modules consume from PyEPO.

The reference subclasses ``pyepo.func.abcmodule.optModule`` and reads
``self.optmodel.modelSense``, ``self.processes``, ``self.pool``,
``self._branch_rng`` and ``self._reduce`` from it (/root/reference
src/cave.py:53,62-67,73,126,195,201).  PyEPO is not part of /root/reference and
is not installed here, so when it is importable we use the real classes, and
otherwise this minimal stand-in that provides exactly those attributes
(``reduction`` semantics as documented in the reference README.md:75,90).
"""

from __future__ import annotations

from enum import Enum

import numpy as np
from torch import nn

try:  # real PyEPO, when present
    from pyepo import EPO  # type: ignore
    from pyepo.func.abcmodule import optModule  # type: ignore

    HAS_PYEPO = True
except Exception:  # noqa: BLE001 - PyEPO absent (or broken): use the stand-in
    HAS_PYEPO = False

    class EPO(Enum):
        """pyepo.EPO: sense of the optimisation model."""
        MINIMIZE = 1
        MAXIMIZE = -1

    class optModule(nn.Module):
        """Stand-in for pyepo.func.abcmodule.optModule (attributes used by CaVE only)."""

        def __init__(self, optmodel, processes: int = 1, solve_ratio: float = 1.0, reduction: str = "mean",
                     dataset=None) -> None:
            super().__init__()
            if not hasattr(optmodel, "modelSense"):
                raise TypeError("arg model is not an optModel")
            if processes < 0:
                raise ValueError(f"Invalid processes {processes}.")
            if reduction not in ("mean", "sum", "none"):
                raise ValueError(f"No reduction '{reduction}'.")
            self.optmodel = optmodel
            self.processes = processes
            self.pool = None  # the pathos worker pool is a CPU-path concern; solver='hip' never uses it
            self.solve_ratio = solve_ratio
            self.reduction = reduction
            self.dataset = dataset
            self._branch_rng = np.random.RandomState()

        def _reduce(self, loss):
            if self.reduction == "mean":
                return loss.mean()
            if self.reduction == "sum":
                return loss.sum()
            return loss


def sense_sign(model_sense) -> float:
    """-1 for MINIMIZE, +1 for MAXIMIZE, else ValueError (src/cave.py:62-67)."""
    if model_sense == EPO.MINIMIZE or model_sense == EPO.MINIMIZE.value:
        return -1.0
    if model_sense == EPO.MAXIMIZE or model_sense == EPO.MAXIMIZE.value:
        return 1.0
    raise ValueError("Invalid modelSense. Must be EPO.MINIMIZE or EPO.MAXIMIZE.")
