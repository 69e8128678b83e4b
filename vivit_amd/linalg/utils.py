"""Helpers of ``vivit_amd.linalg`` (mirror of vivit/linalg/utils.py)."""
from typing import Callable, Dict, List, Union

from torch import Tensor
from torch.nn import Module

from vivit_amd import kernels


def real_backpack_extensions():
    """``backpack.extensions`` if backpack-for-pytorch is installed, else ``None``."""
    try:
        import backpack.extensions as ext  # noqa: WPS433  (third-party, optional)

        return ext
    except Exception:
        return None


def get_vivit_extension(subsampling: Union[None, List[int]], mc_samples: int):
    """Extension that provides the Gram / ``V`` closures for a backward pass.

    Without BackPACK: ``ViViTGGN{Exact,MC}`` of the stand-in backend (factorised Linear weights,
    vivit/linalg/utils.py:11-28).  With BackPACK installed: its ``SqrtGGN{Exact,MC}`` with the Linear module
    extension replaced by one that keeps the weight factorised (``extensions/backpack_adapter.py``; untestable here,
    falls back to the plain extension on any mismatch); the hooks wrap materialised ``V_t`` tensors into the same
    closures (see ``get_closures``).
    """
    ext = real_backpack_extensions()
    if ext is not None:
        from vivit_amd.extensions.backpack_adapter import factorised_sqrt_ggn

        fact = factorised_sqrt_ggn(mc_samples, subsampling)   # Linear weights as closures (linear.py:41-81); None: not possible
        if fact is not None:
            return fact
        if mc_samples == 0:
            return ext.SqrtGGNExact(subsampling=subsampling)
        return ext.SqrtGGNMC(mc_samples=mc_samples, subsampling=subsampling)
    from vivit_amd.backend.extensions import ViViTGGNExact, ViViTGGNMC

    if mc_samples == 0:
        return ViViTGGNExact(subsampling=subsampling)
    return ViViTGGNMC(mc_samples=mc_samples, subsampling=subsampling)


def get_closures(param, savefield: str) -> Dict[str, Callable]:
    """``{"gram_mat", "V_mat_prod", "V_t_mat_prod"}`` for ``param`` whatever provided the factor."""
    value = getattr(param, savefield)
    if isinstance(value, dict):
        return value
    from vivit_amd.backend.extensions import _materialised_closures

    return _materialised_closures(value)


def use_parameter_side(params, n: int, side: str) -> bool:
    """``True`` if the group's eigenproblem should be solved on the parameter side.

    The GGN block ``V^T V`` (``P x P``) and the Gram matrix ``V V^T`` (``n x n``) share their non-zero
    spectrum; the reference always decomposes the Gram matrix (vivit/linalg/eigvalsh.py:215-225).  When a
    group has fewer parameters than Gram rows (LeNet's conv1: P = 456 against n = N*C = 20 480) the
    ``P x P`` problem is the cheap one and its eigenvectors already live in parameter space.
    ``side``: ``"auto"`` (smaller side), ``"gram"`` (reference behaviour), ``"param"``.
    """
    if side not in ("auto", "gram", "param"):
        raise ValueError(f"side must be 'auto', 'gram' or 'param', got {side!r}")
    if side != "auto":
        return side == "param"
    return sum(int(p.numel()) for p in params) < n


def parameter_side_symeig(params, savefield: str, eigenvectors: bool):
    """Eigen-decomposition of the group's ``P x P`` GGN block ``H = V^T V``, padded to the Gram spectrum.

    Returns ``(evals [n] ascending, Q [P, P] or None, cols [n])``: ``n - P`` entries of ``evals`` are the exact
    zeros the (rank-deficient) Gram matrix has in their place -- inserted where they belong in the ascending order,
    i.e. after any negative rounding noise of ``H``'s own spectrum; ``cols[i]`` is the column of ``Q`` holding the
    unit parameter-space eigenvector of ``evals[i]`` (concatenated over ``params`` in order) or ``-1`` for a padded
    zero, which has no GGN eigenvector.  If ``P >= n`` the top ``n`` eigenvalues are returned (the remaining
    ``P - n`` are zero up to rounding).
    """
    import torch

    facs = []
    for prm in params:
        c = get_closures(prm, savefield)
        C, N = c["shape_cn"]
        facs.append(c["factor"]().reshape(C * N, -1))
    n = facs[0].shape[0]
    Vcat = facs[0] if len(facs) == 1 else torch.cat(facs, dim=1)  # [n, P]
    P = Vcat.shape[1]
    H = kernels.gemm_tn(Vcat, Vcat)                                 # [P, P] = V^T V, contraction over the n rows
    w, Q = kernels.symeig(H, eigenvectors=eigenvectors, overwrite=True)
    if P >= n:
        cols = torch.arange(P - n, P, device=w.device)
        return w[P - n:], (Q if eigenvectors else None), cols
    neg = int((w < 0).sum().item())  # rounding noise below zero sorts in front of the exact zeros
    zeros = torch.zeros(n - P, dtype=w.dtype, device=w.device)
    evals = torch.cat([w[:neg], zeros, w[neg:]])
    cols = torch.cat([
        torch.arange(neg, device=w.device),
        torch.full((n - P,), -1, dtype=torch.long, device=w.device),
        torch.arange(neg, P, device=w.device),
    ])
    return evals, Q, cols


def get_hook_store_batch_size(
    param_groups: List[Dict], destination: Dict[int, int], verbose: bool = False
) -> Callable[[Module], None]:
    """Hook that records the batch size once, for every group (vivit/linalg/utils.py:31-64)."""

    def hook_store_batch_size(module: Module):
        if destination == {}:
            batch_size = module.input0.shape[0]
            for group in param_groups:
                if verbose:
                    print(f"Group {id(group)}: Store 'batch_size'")
                destination[id(group)] = batch_size

    return hook_store_batch_size


def normalize(tensors: List[Tensor]):
    """Unit-normalise stacked vectors in parameter-list format, in place (K10;
    vivit/linalg/utils.py:67-76): one fused sum-of-squares pass + one scaling pass per tensor."""
    for idx in range(len(tensors)):
        if not tensors[idx].is_contiguous():
            tensors[idx] = tensors[idx].contiguous()
    kernels.normalize_rows_(tensors)
