"""vivit_amd: the low-rank GGN curvature path of ViViT on AMD Instinct MI355X.

Public API (same names as ``vivit``, vivit/__init__.py:4-17).  All tensor work runs in the
hand-written gfx950 kernels of ``libvivit_hip.so`` (see ``include/vivit_hip.h``); importing the
package does not need a GPU, computing anything does.
"""
from vivit_amd import extensions, optim
from vivit_amd.linalg.eigh import EighComputation
from vivit_amd.linalg.eigvalsh import EigvalshComputation
from vivit_amd.optim.directional_damped_newton import DirectionalDampedNewtonComputation
from vivit_amd.optim.directional_derivatives import DirectionalDerivativesComputation

__version__ = "0.1.0"

__all__ = [
    "extensions",
    "optim",
    "EigvalshComputation",
    "EighComputation",
    "DirectionalDerivativesComputation",
    "DirectionalDampedNewtonComputation",
]
