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


def _is_factorised(t) -> bool:
    from vivit_amd.backend.extensions import LinearFactor

    return isinstance(t, LinearFactor)


def _dense(t):
    return t.materialise() if _is_factorised(t) else t


def gram_of_factor(V, out=None, beta: float = 0.0):
    """``V^T V`` as ``[C, N, C, N]``: K1 (SYRK) for a materialised ``V``; for a factorised Linear weight
    ``(z z^T) o (s s^T)`` -- two small SYRKs and the fused Hadamard kernel (K1',
    vivit/extensions/secondorder/vivit/linear.py:72-75)."""
    if not _is_factorised(V):
        return partial_contract(V, V, start_dims=(2, 2), out=out, beta=beta)
    C, N, O = V.s.shape
    Gz = kernels.gram_syrk(V.z)
    Gs = kernels.gram_syrk(V.s.reshape(C * N, O))
    G = kernels.gram_hadamard(Gz, Gs, C, N, out=None if out is None else out.view(C * N, C * N), alpha=1.0, beta=beta)
    return G.view(C, N, C, N)


def factor_dot_grad(V, g, out=None, beta: float = 0.0):
    """``V^T g`` as ``[C, N_ggn, N_grad]`` (K2, vivit/optim/directional_damped_newton.py:255).  Both factorised:
    ``sum_{o,i} s[c,n,o] z[n,i] delta[m,o] zg[m,i] = (z_n . zg_m) (s_cn . delta_m)``, two NT GEMMs + Hadamard."""
    if _is_factorised(V) and _is_factorised(g):
        C, N, O = V.s.shape
        M = g.s.shape[0]
        Gz = kernels.gemm_nt(V.z, g.z)                                 # [N, M]
        Gs = kernels.gemm_nt(V.s.reshape(C * N, O), g.s)               # [C N, M]
        R = kernels.gram_hadamard_block(Gz, Gs, C, N, 1, M, out=None if out is None else out.view(C * N, M), alpha=1.0,
                                        beta=beta)
        return R.view(C, N, M)
    return partial_contract(_dense(V), _dense(g), start_dims=(2, 1), out=out, beta=beta)


def apply_factor(V, coef):
    """``sum_{c,n} coef[k,c,n] V[c,n,...]`` -> ``[K, *param]`` (K7/K8): streaming GEMM over a materialised ``V``;
    for a factorised Linear weight the length-C contraction kernel followed by one GEMM with ``z``
    (vivit/extensions/secondorder/vivit/linear.py:53)."""
    K = coef.shape[0]
    if _is_factorised(V):
        C, N, O = V.s.shape
        T = kernels.class_contract(coef.reshape(K, C, N).contiguous(), V.s)    # [K, O, N]
        return kernels.gemm_nn(T.view(K * O, N), V.z).view(K, O, V.z.shape[1])
    Vd = V.detach()
    n = Vd.shape[0] * Vd.shape[1]
    return kernels.gemm_nn(coef.reshape(K, n).contiguous(), Vd.reshape(n, -1)).view(K, *Vd.shape[2:])


DP_ROWS_MAX_COLUMNS = 4096  # data parallel: materialised factors narrower than this are all-gathered (block rows)


def dot_products(hook: ParameterGroupsHook, param, savefield_ggn: str, savefield_grad: str, verbose: bool,
                 data_parallel=None):
    """``V^T V`` (K1) and ``V^T g`` (K2) of one parameter, accumulated in-kernel (beta = 1) into
    the group's running sums when they exist (vivit/optim/directional_derivatives.py:216-252).

    ``data_parallel``: ``None`` or ``{"group": process_group}`` -- the factors then hold this rank's batch shard only
    and the sums are built by :class:`vivit_amd.distributed.BatchShardedGram` (collectives inside)."""
    V = getattr(param, savefield_ggn)
    g = getattr(param, savefield_grad)
    if verbose:
        print(f"Param {id(param)}: Compute V_t_V and V_t_g_n")
    existing = hook.current_accumulation(param)
    if data_parallel is not None:
        from vivit_amd.distributed import BatchShardedGram

        if existing is None:
            existing = {"dp_acc": BatchShardedGram(V.shape[0], V.shape[1], data_parallel.get("group"),
                                                   N_grad_local=g.shape[0])}
        acc = existing["dp_acc"]
        if _is_factorised(V) and _is_factorised(g):
            acc.add_linear(V.s, V.z, g.s, g.z)
        else:
            Vd, gd = _dense(V).detach(), _dense(g).detach()
            if Vd[0, 0].numel() < DP_ROWS_MAX_COLUMNS:
                acc.add_factor_rows(Vd, gd)
            else:
                acc.add_factor(Vd, gd)
        return existing
    if existing is None:
        return {"V_t_V": gram_of_factor(V), "V_t_g_n": factor_dot_grad(V, g)}
    gram_of_factor(V, out=existing["V_t_V"], beta=1.0)
    factor_dot_grad(V, g, out=existing["V_t_g_n"], beta=1.0)
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
    Returns ``(evals[K], evecs[n,K], gammas[N_grad,K], lambdas[N_ggn,K], V_correction, C, N_ggn, dp_acc)``
    (``dp_acc``: the :class:`BatchShardedGram` of a data-parallel run, else ``None``; all sizes are global then).
    """
    group_id = id(group)
    dp_acc = accumulation.pop("dp_acc", None)
    if dp_acc is not None:
        # data parallel: assemble the batch-sharded sums (collectives).  ``N`` stays the LOCAL batch size -- it only
        # compensates the 1/sqrt(N) and 1/N that this rank's backward pass put into V and g -- while N_ggn below
        # becomes the global number of curvature samples, so V_correction = sqrt(N_local / N_ggn_global) turns the
        # sums into those of the mean loss over the global batch.
        V_t_V = dp_acc.finalize()
        accumulation["V_t_g_n"] = dp_acc.finalize_vtg()
    else:
        V_t_V = accumulation.pop("V_t_V")
    C, N_ggn = V_t_V.shape[0], V_t_V.shape[1]
    V_correction = math.sqrt(N / N_ggn)  # compensates BackPACK's 1/sqrt(N) and the sub-sampling
    gram_unscaled = reshape_as_square(V_t_V)

    if verbose:
        print(f"Group {group_id}: Eigen-decompose Gram matrix")
    # reduction + all eigenvalues, criterion on the host, then only the kept eigenvectors (the reference computes all
    # n eigenvectors and slices: directional_damped_newton.py:315-321)
    plan = kernels.symeig_reduce(gram_unscaled)
    evals = plan.evals
    evals *= V_correction**2

    keep = group["criterion"](evals)
    if verbose:
        print(f"Group {group_id}: Filter directions ({len(evals)} → {len(keep)})")
    evals, evecs = evals[keep], plan.select(keep).contiguous()
    del plan

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
    return evals, evecs, gammas, lambdas, V_correction, C, N_ggn, dp_acc


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
        factorised: bool = False,
        data_parallel: bool = False,
        process_group=None,
    ):
        """``factorised``, ``data_parallel``, ``process_group`` are not in the reference: see
        :class:`vivit_amd.optim.DirectionalDampedNewtonComputation`."""
        check_subsampling_unique(subsampling_grad)
        check_subsampling_unique(subsampling_ggn)
        self._mc_samples_ggn = mc_samples_ggn
        if self._mc_samples_ggn != 0:
            assert mc_samples_ggn == 1
        self._subsampling_grad = subsampling_grad
        self._subsampling_ggn = subsampling_ggn
        self._factorised = factorised
        self._dp = {"group": process_group} if data_parallel else None
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
            get_batch_grad_extension(self._subsampling_grad, factorised=self._factorised),
            get_sqrt_ggn_extension(subsampling=self._subsampling_ggn, mc_samples=self._mc_samples_ggn,
                                   factorised=self._factorised),
        ]

    def get_extension_hook(self, param_groups: List[Dict]) -> Callable[[Module], None]:
        self._check_param_groups(param_groups)
        store_batch_size = get_hook_store_batch_size(param_groups, self._batch_size, verbose=self._verbose)
        hook = ParameterGroupsHook.from_functions(
            param_groups,
            lambda hook, param: self._param_computation(
                hook, param, self._savefield_ggn, self._savefield_grad, self._verbose, self._dp
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
    def _param_computation(hook, param, savefield_ggn, savefield_grad, verbose, data_parallel=None):
        result = dot_products(hook, param, savefield_ggn, savefield_grad, verbose, data_parallel)
        # neither factor is needed again (directional_derivatives.py:249-250)
        delete_savefield(param, savefield_ggn, verbose=verbose)
        delete_savefield(param, savefield_grad, verbose=verbose)
        return result

    @staticmethod
    def _group_hook(hook, accumulation, group, batch_size, gammas, lambdas, verbose, warn_small_eigvals):
        N = batch_size.pop(id(group))
        _, _, gam, lam, _, _, _, _ = gram_space_directions(
            accumulation, group, N, verbose, warn_small_eigvals, _SMALL_EVALS_GAMMA
        )
        gammas[id(group)] = gam
        lambdas[id(group)] = lam

    @staticmethod
    def _check_param_groups(param_groups: List[Dict]):
        check_key_exists(param_groups, "params")
        check_key_exists(param_groups, "criterion")
        check_unique_params(param_groups)
