"""1st- and 2nd-order directional derivatives along GGN eigenvectors.

API of ``vivit.optim.directional_derivatives`` (vivit/optim/directional_derivatives.py:24-364),
computed on the MI355X kernels.  The Gram-space part shared with the damped Newton step lives
in :func:`gram_space_directions`.
"""
import math
from typing import Callable, Dict, List, Optional, Tuple
from warnings import warn

from torch import Tensor
from torch.nn import Module

from vivit_amd import kernels
from vivit_amd.linalg.utils import get_hook_store_batch_size
from vivit_amd.optim.utils import get_batch_grad_extension, get_sqrt_ggn_extension
from vivit_amd.utils import delete_savefield
from vivit_amd.utils.checks import check_key_exists, check_subsampling_unique, check_unique_params
from vivit_amd.utils.gram import partial_contract, reshape_as_square
from vivit_amd.utils.hooks import ParameterGroupsHook

_SMALL_EVALS_GAMMA = (
    "Some eigenvalues are small. This can lead to numerical instabilities"
    + " in the directional gradients because they require division by the"
    + " eigenvalue square root."
    + " Maybe use a more restrictive eigenvalue filter criterion."
)


def dot_products(hook: ParameterGroupsHook, param, savefield_ggn: str, savefield_grad: str, verbose: bool):
    """``V^T V`` (K1) and ``V^T g`` (K2) of one parameter, accumulated in-kernel (beta = 1) into
    the group's running sums when they exist (vivit/optim/directional_derivatives.py:216-252)."""
    V = getattr(param, savefield_ggn)
    g = getattr(param, savefield_grad)
    if verbose:
        print(f"Param {id(param)}: Compute V_t_V and V_t_g_n")
    existing = hook.current_accumulation(param)
    if existing is None:
        return {
            "V_t_V": partial_contract(V, V, start_dims=(2, 2)),
            "V_t_g_n": partial_contract(V, g, start_dims=(2, 1)),
        }
    partial_contract(V, V, start_dims=(2, 2), out=existing["V_t_V"], beta=1.0)
    partial_contract(V, g, start_dims=(2, 1), out=existing["V_t_g_n"], beta=1.0)
    return existing


def accumulate_dot_products(existing: Dict[str, Tensor], update: Dict[str, Tensor], verbose: bool):
    """``existing += update`` unless the kernel already did it (same dict object)."""
    if update is existing:
        return existing
    for key in existing.keys():
        if verbose:
            print(f"Accumulate dot product {key}")
        existing[key].add_(update[key])
    return existing


def gram_space_directions(accumulation: Dict[str, Tensor], group: Dict, N: int, verbose: bool,
                          warn_small_eigvals: float, warning: str):
    """Eigen-decompose the Gram matrix, filter directions, evaluate gammas and lambdas.

    Follows vivit/optim/directional_damped_newton.py:304-351 line by line; scalar factors are
    folded into kernel ``alpha``s instead of separate elementwise passes:
      gram = Vc^2 V_t_V is never formed -- eigenvalues are scaled (O(n)) and ``alpha = Vc^2`` is
      applied inside the ``G @ E`` GEMM.
    Returns ``(evals[K], evecs[n,K], gammas[N_grad,K], lambdas[N_ggn,K], V_correction, C, N_ggn)``.
    """
    group_id = id(group)
    V_t_V = accumulation.pop("V_t_V")
    C, N_ggn = V_t_V.shape[0], V_t_V.shape[1]
    V_correction = math.sqrt(N / N_ggn)  # compensates BackPACK's 1/sqrt(N) and the sub-sampling
    gram_unscaled = reshape_as_square(V_t_V)

    if verbose:
        print(f"Group {group_id}: Eigen-decompose Gram matrix")
    evals, evecs = kernels.symeig(gram_unscaled, eigenvectors=True)
    evals *= V_correction**2

    keep = group["criterion"](evals)
    if verbose:
        print(f"Group {group_id}: Filter directions ({len(evals)} → {len(keep)})")
    evals, evecs = evals[keep], evecs[:, keep].contiguous()

    if verbose:
        print(f"Group {group_id}: Compute gammas")
    V_t_g_n = accumulation.pop("V_t_g_n").flatten(start_dim=0, end_dim=1)  # [n, N_grad]

    if (evals.abs() < warn_small_eigvals).any():
        warn(warning)

    # gammas[n, d] = sum_i (Vc N V_t_g_n)[i, n] evecs[i, d] / sqrt(evals[d])      (K5)
    gammas = kernels.gemm_tn(V_t_g_n, evecs, alpha=V_correction * N)
    kernels.scale_cols_rsqrt_(gammas, evals)

    if verbose:
        print(f"Group {group_id}: Compute lambdas")
    # lambdas[n, d] = sum_c (sqrt(N_ggn) (gram E)[(c,n), d])^2 / evals[d]         (K6)
    GE = kernels.gemm_nn(gram_unscaled, evecs, alpha=V_correction**2)
    lambdas = kernels.dir_curvature(GE, evals, C, N_ggn, scale=float(N_ggn))
    return evals, evecs, gammas, lambdas, V_correction, C, N_ggn


class DirectionalDerivativesComputation:
    """Provide extensions and the hook for 1st/2nd-order directional derivatives.

    ``get_result(group) -> (gammas [N_grad, K], lambdas [N_ggn, K])``.  Groups need ``'params'``
    and ``'criterion'``.  The loss must use ``reduction='mean'``.
    """

    def __init__(
        self,
        subsampling_grad: Optional[List[int]] = None,
        subsampling_ggn: Optional[List[int]] = None,
        mc_samples_ggn: Optional[int] = 0,
        verbose: Optional[bool] = False,
        warn_small_eigvals: float = 1e-4,
    ):
        check_subsampling_unique(subsampling_grad)
        check_subsampling_unique(subsampling_ggn)
        self._mc_samples_ggn = mc_samples_ggn
        if self._mc_samples_ggn != 0:
            assert mc_samples_ggn == 1
        self._subsampling_grad = subsampling_grad
        self._subsampling_ggn = subsampling_ggn
        self._savefield_grad = get_batch_grad_extension(None).savefield
        self._savefield_ggn = get_sqrt_ggn_extension(None, mc_samples_ggn).savefield
        self._verbose = verbose
        self._warn_small_eigvals = warn_small_eigvals
        self._batch_size: Dict[int, int] = {}
        self._gammas: Dict[int, Tensor] = {}
        self._lambdas: Dict[int, Tensor] = {}

    def get_result(self, group: Dict) -> Tuple[Tensor, Tensor]:
        try:
            return self._gammas[id(group)], self._lambdas[id(group)]
        except KeyError as e:
            raise KeyError("No results available for this group") from e

    def get_extensions(self) -> List:
        return [
            get_batch_grad_extension(self._subsampling_grad),
            get_sqrt_ggn_extension(subsampling=self._subsampling_ggn, mc_samples=self._mc_samples_ggn),
        ]

    def get_extension_hook(self, param_groups: List[Dict]) -> Callable[[Module], None]:
        self._check_param_groups(param_groups)
        store_batch_size = get_hook_store_batch_size(param_groups, self._batch_size, verbose=self._verbose)
        hook = ParameterGroupsHook.from_functions(
            param_groups,
            lambda hook, param: self._param_computation(
                hook, param, self._savefield_ggn, self._savefield_grad, self._verbose
            ),
            lambda hook, accumulation, group: self._group_hook(
                hook, accumulation, group, self._batch_size, self._gammas, self._lambdas, self._verbose,
                self._warn_small_eigvals,
            ),
            lambda hook, existing, update: accumulate_dot_products(existing, update, self._verbose),
        )

        def extension_hook(module: Module):
            if self._verbose:
                print(f"Extension hook on module {id(module)} {module}")
            store_batch_size(module)
            hook(module)

        if self._verbose:
            print("ID map groups → params")
            for group in param_groups:
                print(f"{id(group)} → {[id(p) for p in group['params']]}")
        return extension_hook

    @staticmethod
    def _param_computation(hook, param, savefield_ggn, savefield_grad, verbose):
        result = dot_products(hook, param, savefield_ggn, savefield_grad, verbose)
        # neither factor is needed again (directional_derivatives.py:249-250)
        delete_savefield(param, savefield_ggn, verbose=verbose)
        delete_savefield(param, savefield_grad, verbose=verbose)
        return result

    @staticmethod
    def _group_hook(hook, accumulation, group, batch_size, gammas, lambdas, verbose, warn_small_eigvals):
        N = batch_size.pop(id(group))
        _, _, gam, lam, _, _, _ = gram_space_directions(
            accumulation, group, N, verbose, warn_small_eigvals, _SMALL_EVALS_GAMMA
        )
        gammas[id(group)] = gam
        lambdas[id(group)] = lam

    @staticmethod
    def _check_param_groups(param_groups: List[Dict]):
        check_key_exists(param_groups, "params")
        check_key_exists(param_groups, "criterion")
        check_unique_params(param_groups)
