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
    vivit/linalg/utils.py:11-28).  With BackPACK installed: its built-in ``SqrtGGN{Exact,MC}``;
    the hooks then wrap the materialised ``V_t`` into the same closures (see ``get_closures``).
    """
    ext = real_backpack_extensions()
    if ext is not None:
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
