"""Eigensolver wrappers with the contract of ``vivit.utils.eig`` on the HIP solver.

``Tensor.symeig`` (removed from torch) is replaced by ``vivit_symeig_f32``: ascending
eigenvalues, eigenvectors column-wise.
"""
import torch

from vivit_amd import kernels


def _device_symeig(mat, eigenvectors, upper):
    # The kernel reads the lower triangle; ``upper=True`` (the old default) reads the upper one,
    # i.e. the lower triangle of the transpose.
    # The solver destroys its input, the reference never mutates the caller's matrix: always work on a private
    # row-major copy (``.contiguous()`` alone would alias a column-major input such as ``G.T``).
    src = mat.detach()
    work = (src.t() if upper else src).clone(memory_format=torch.contiguous_format)
    evals, evecs = kernels.symeig(work, eigenvectors=eigenvectors, overwrite=True)
    if evecs is None:
        evecs = mat.new_empty(0)
    return evals, evecs


def _has_nans(tensor):
    return torch.any(torch.isnan(tensor))


def shift_diag(input, shift, inplace=False):
    """Add ``shift`` to the diagonal of a 2d tensor (vivit/utils/eig.py:51-74)."""
    if shift == 0.0:
        return input
    result = input if inplace else input.clone()
    result.diagonal().add_(shift)
    return result


def symeig_psd(input, eigenvectors=False, upper=True, shift=0.0, shift_inplace=False):
    """Eigen-decomposition of a PSD matrix with an optional diagonal shift (eig.py:6-48)."""
    if input.dim() != 2:
        raise ValueError(f"Input must have dimension 2. Got {input.dim()}.")
    input = shift_diag(input, shift, inplace=shift_inplace)
    try:
        evals, evecs = _device_symeig(input, eigenvectors, upper)
    except RuntimeError as e:
        raise RuntimeError(f"Tensor contains NaNs: {_has_nans(input)}") from e
    if shift_inplace:
        input = shift_diag(input, -shift, inplace=shift_inplace)
    evals -= shift
    return evals, evecs


def remove_zero_evals(evals, evecs, atol=1e-7, rtol=1e-5):
    """Drop (eigenvalue, eigenvector) pairs with eigenvalue ~ 0 (eig.py:111-134)."""
    nonzero = torch.isclose(evals, torch.zeros_like(evals), rtol=rtol, atol=atol).logical_not()
    evals = evals[nonzero]
    if evecs.numel() != 0:
        evecs = evecs[:, nonzero]
    return evals, evecs


def symeig(input, eigenvectors=False, upper=True, atol=1e-7, rtol=1e-5):
    """Eigen-decomposition with the numerically-zero pairs removed (eig.py:77-108)."""
    if input.dim() != 2:
        raise ValueError("Input must be of dimension 2")
    try:
        evals, evecs = _device_symeig(input, eigenvectors, upper)
    except RuntimeError as e:
        raise RuntimeError(f"Tensor contains NaNs: {_has_nans(input)}") from e
    return remove_zero_evals(evals, evecs, atol=atol, rtol=rtol)
