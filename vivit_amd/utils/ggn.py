"""Apply the GGN square root ``V`` to stacked Gram-space vectors (API of ``vivit.utils.ggn``)."""
from typing import List

from torch import Tensor, cat

from vivit_amd import kernels


def Vmp(V_t: Tensor, mat: Tensor, start_dim: int) -> Tensor:
    """``V @ mat``: ``[F, *start] x [*start, *param] -> [F, *param]`` (vivit/utils/ggn.py:94-115).

    One NN GEMM ``[F, n] x [n, P]`` that streams ``V_t`` once (K7/K8; HBM-bound for small F).
    """
    n = 1
    for s in V_t.shape[:start_dim]:
        n *= int(s)
    A = mat.detach().reshape(mat.shape[0], n)
    B = V_t.detach().reshape(n, -1)
    return kernels.gemm_nn(A, B).view(mat.shape[0], *V_t.shape[start_dim:])


def _get_V_t(param, savefield: str, subsampling: List[int] = None) -> Tensor:
    V_t = getattr(param, savefield)
    if subsampling is not None:
        V_t = V_t[:, subsampling]
    return V_t


def V_param_mat_prod(param, mat: Tensor, savefield: str, subsampling: List[int] = None) -> Tensor:
    """``V @ mat`` for one parameter's stored ``V_t`` (ggn.py:73-91)."""
    return Vmp(_get_V_t(param, savefield, subsampling=subsampling), mat, 2)


def V_mat_prod(mat: Tensor, parameters, savefield: str, subsampling: List[int] = None, concat: bool = False):
    """``V @ mat`` over a parameter list; ``mat`` must be ``[F, C, N]`` (ggn.py:11-50)."""
    assert mat.dim() == 3, f"mat must be [F, C, N]. Got {mat.dim()} dimensions."
    result = [V_param_mat_prod(p, mat, savefield, subsampling=subsampling) for p in parameters]
    if concat:
        result = cat([r.flatten(start_dim=1) for r in result], dim=1)
    return result
