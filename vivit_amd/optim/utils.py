"""Helpers of ``vivit_amd.optim`` (mirror of vivit/optim/utils.py)."""
from typing import List, Union

from vivit_amd.linalg.utils import real_backpack_extensions


def get_sqrt_ggn_extension(subsampling: Union[None, List[int]], mc_samples: int):
    """``SqrtGGN{Exact,MC}`` (BackPACK's if installed, else the stand-in backend's);
    vivit/optim/utils.py:8-25."""
    ext = real_backpack_extensions()
    if ext is None:
        from vivit_amd.backend import extensions as ext
    if mc_samples == 0:
        return ext.SqrtGGNExact(subsampling=subsampling)
    return ext.SqrtGGNMC(subsampling=subsampling, mc_samples=mc_samples)


def get_batch_grad_extension(subsampling: Union[None, List[int]]):
    ext = real_backpack_extensions()
    if ext is None:
        from vivit_amd.backend import extensions as ext
    return ext.BatchGrad(subsampling=subsampling)
