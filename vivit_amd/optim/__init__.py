"""Optimisation methods on the low-rank GGN (mirror of ``vivit.optim``)."""
from vivit_amd.optim.directional_damped_newton import DirectionalDampedNewtonComputation
from vivit_amd.optim.directional_derivatives import DirectionalDerivativesComputation
from vivit_amd.optim.optimizer import DirectionalDampedNewton

__all__ = ["DirectionalDerivativesComputation", "DirectionalDampedNewtonComputation", "DirectionalDampedNewton"]
