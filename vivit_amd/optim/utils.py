"""Helpers of ``vivit_amd.optim`` (mirror of vivit/optim/utils.py)."""
from typing import List, Union

from vivit_amd.linalg.utils import real_backpack_extensions


def _pick_backend(factorised: bool):
    """BackPACK's extension module if it is installed, else the stand-in backend's.  ``factorised=True`` exists only in the
    stand-in backend: under BackPACK's own ``backpack(...)`` context a stand-in extension is not a ``BackpropExtension`` (it
    would fail deep inside ``backward`` or be ignored), so that combination is refused here."""
    ext = real_backpack_extensions()
    if ext is not None and factorised:
        raise NotImplementedError(
            "factorised=True needs the stand-in backend (`from vivit_amd.backend import backpack, extend`): BackPACK's "
            "SqrtGGN{Exact,MC} / BatchGrad materialise Linear weights.  Use factorised=False with BackPACK installed.")
    if ext is None:
        from vivit_amd.backend import extensions as ext
    return ext


def get_sqrt_ggn_extension(subsampling: Union[None, List[int]], mc_samples: int, factorised: bool = False):
    """``SqrtGGN{Exact,MC}`` (BackPACK's if installed, else the stand-in backend's);
    vivit/optim/utils.py:8-25.  ``factorised=True`` (stand-in backend only): Linear weights keep the factorised
    form ``(s, z)`` of vivit/extensions/secondorder/vivit/linear.py:41-42 instead of the materialised tensor."""
    ext = _pick_backend(factorised)
    kw = {"factorised": True} if factorised else {}
    if mc_samples == 0:
        return ext.SqrtGGNExact(subsampling=subsampling, **kw)
    return ext.SqrtGGNMC(subsampling=subsampling, mc_samples=mc_samples, **kw)


def get_batch_grad_extension(subsampling: Union[None, List[int]], factorised: bool = False):
    ext = _pick_backend(factorised)
    kw = {"factorised": True} if factorised else {}
    return ext.BatchGrad(subsampling=subsampling, **kw)
