"""Helpers of ``vivit_amd.optim`` (mirror of vivit/optim/utils.py)."""
from typing import List, Union

from vivit_amd.linalg.utils import real_backpack_extensions


def get_sqrt_ggn_extension(subsampling: Union[None, List[int]], mc_samples: int, factorised: bool = False):
    """``SqrtGGN{Exact,MC}`` (BackPACK's if installed, else the stand-in backend's);
    vivit/optim/utils.py:8-25.  ``factorised=True`` (stand-in backend only): Linear weights keep the factorised
    form ``(s, z)`` of vivit/extensions/secondorder/vivit/linear.py:41-42 instead of the materialised tensor."""
    ext = real_backpack_extensions()
    if ext is None or factorised:
        from vivit_amd.backend import extensions as ext
    kw = {"factorised": True} if factorised else {}
    if mc_samples == 0:
        return ext.SqrtGGNExact(subsampling=subsampling, **kw)
    return ext.SqrtGGNMC(subsampling=subsampling, mc_samples=mc_samples, **kw)


def get_batch_grad_extension(subsampling: Union[None, List[int]], factorised: bool = False):
    ext = real_backpack_extensions()
    if ext is None or factorised:
        from vivit_amd.backend import extensions as ext
    kw = {"factorised": True} if factorised else {}
    return ext.BatchGrad(subsampling=subsampling, **kw)
