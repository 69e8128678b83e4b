"""GGN eigenvalues during backpropagation (API of ``vivit.linalg.eigvalsh``)."""
from typing import Any, Callable, Dict, List

from torch import Tensor
from torch.nn import Module, Parameter

from vivit_amd import kernels
from vivit_amd.linalg.utils import (
    get_closures,
    get_hook_store_batch_size,
    get_vivit_extension,
    parameter_side_symeig,
    use_parameter_side,
)
from vivit_amd.utils import delete_savefield
from vivit_amd.utils.checks import check_key_exists, check_subsampling_unique, check_unique_params
from vivit_amd.utils.gram import reshape_as_square
from vivit_amd.utils.hooks import ParameterGroupsHook


class EigvalshComputation:
    """Provide the extension and the extension hook that compute GGN eigenvalues.

    Same constructor, methods, parameter-group keys and error conventions as
    vivit/linalg/eigvalsh.py:20-237.  The loss must use ``reduction='mean'``.
    Data flow on the device: per parameter one SYRK (or the factorised Linear path) accumulated
    in place into the group's ``[n, n]`` Gram (beta = 1), then one values-only ``symeig``.
    """

    def __init__(self, subsampling: List[int] = None, mc_samples: int = 0, verbose: bool = False, side: str = "gram",
                 data_parallel: bool = False, process_group=None):
        """``side`` (not in the reference): ``"auto"`` solves a group on its parameter side (``P x P``) when it has
        fewer parameters than Gram rows, ``"gram"`` always decomposes the Gram matrix like the reference.
        ``data_parallel`` / ``process_group`` (not in the reference): every rank back-propagated ITS batch shard; the
        Gram matrix of the global batch is assembled across ranks (vivit_amd.distributed.BatchShardedGram), the
        eigenvalues are those of the GGN of the mean loss over the global batch, identical on every rank."""
        check_subsampling_unique(subsampling)
        use_parameter_side([], 1, side)  # validates ``side``
        if data_parallel and side != "gram":
            raise ValueError("data_parallel needs side='gram'")
        self._dp = {"group": process_group} if data_parallel else None
        self._side = side
        self._subsampling = subsampling
        self._mc_samples = mc_samples
        self._verbose = verbose
        self._savefield = self.get_extension().savefield
        # filled by side effect during backpropagation, keyed by id(group)
        self._batch_size: Dict[int, int] = {}
        self._evals: Dict[int, Tensor] = {}

    def get_result(self, group: Dict) -> Tensor:
        """Eigenvalues (ascending) of the group's GGN block; KeyError if unavailable."""
        try:
            return self._evals[id(group)]
        except KeyError as e:
            raise KeyError("No results available for this group") from e

    def get_extension(self):
        """Extension to pass to ``with backpack(...)``."""
        return get_vivit_extension(self._subsampling, self._mc_samples)

    def get_extension_hook(self, param_groups: List[Dict]) -> Callable[[Module], None]:
        """Hook to pass as ``extension_hook``; groups need the ``'params'`` key."""
        self._check_param_groups(param_groups)
        store_batch_size = get_hook_store_batch_size(param_groups, self._batch_size, verbose=self._verbose)
        hook = ParameterGroupsHook.from_functions(
            param_groups, self.get_param_computation(), self.get_group_hook(), self.get_accumulate()
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

    def get_param_computation(self) -> Callable[[ParameterGroupsHook, Parameter], Tensor]:
        verbose, savefield, side, dp = self._verbose, self._savefield, self._side, self._dp

        def param_computation(self: ParameterGroupsHook, param: Parameter):
            """Gram of this parameter, added in-kernel to the group accumulator if there is one.

            Groups solved on the parameter side keep their factors alive instead (a list of parameters is
            accumulated) until the group hook forms the ``P x P`` block."""
            closures = get_closures(param, savefield)
            group = self.get_group(self._param_to_group[id(param)])
            C, N = closures["shape_cn"]
            if use_parameter_side(group["params"], C * N, side):
                return [param]
            gram_fn = closures["gram_mat"]
            existing = self.current_accumulation(param)
            if dp is not None:  # batch-sharded factors: block rows / parameter shards, assembled in the group hook
                from vivit_amd.distributed import BatchShardedGram

                acc = existing if existing is not None else BatchShardedGram(C, N, dp["group"])
                closures["dp_add"](acc)
                delete_savefield(param, savefield, verbose=verbose)
                return acc
            if existing is None:
                gram = gram_fn()
            else:
                gram_fn(out=existing, beta=1.0)
                gram = existing  # same object: tells ``accumulate`` the sum is already done
            delete_savefield(param, savefield, verbose=verbose)
            return gram

        return param_computation

    def get_accumulate(self) -> Callable[[ParameterGroupsHook, Tensor, Tensor], Tensor]:
        def accumulate(self: ParameterGroupsHook, existing, update):
            if isinstance(existing, list):  # parameter-side group: collect the parameters
                return existing + update
            # ``update`` already is ``existing`` (+= done by the kernel's beta = 1) on the fused path
            if update is existing:
                return update
            return existing.add_(update)

        return accumulate

    def get_group_hook(self) -> Callable[[ParameterGroupsHook, Tensor, Dict[str, Any]], None]:
        batch_sizes, subsampling = self._batch_size, self._subsampling
        evals, verbose, savefield = self._evals, self._verbose, self._savefield

        def group_hook(self: ParameterGroupsHook, accumulation: Tensor, group: Dict):
            group_id = id(group)
            if verbose:
                print(f"Group {group_id}: Delete 'batch_size'")
            batch_size = batch_sizes.pop(group_id)
            scale = None if subsampling is None else batch_size / len(subsampling)  # eigvalsh.py:217-219
            if isinstance(accumulation, list):  # parameter side: P x P block of the GGN, zero-padded spectrum
                gram_evals, _, _ = parameter_side_symeig(group["params"], savefield, eigenvectors=False)
                for param in accumulation:
                    delete_savefield(param, savefield, verbose=verbose)
            else:
                if not isinstance(accumulation, Tensor):  # data-parallel accumulator: assemble (collectives inside)
                    # each rank's factors carry 1/sqrt(N_local); N_local / N_ggn(global) turns the sum into the GGN of
                    # the mean loss over the global batch (covers sub-sampling: len(subsampling) x ranks = N_ggn)
                    scale = batch_size / accumulation.N
                    accumulation = accumulation.finalize()
                gram_mat = reshape_as_square(accumulation)
                gram_evals, _ = kernels.symeig(gram_mat, eigenvectors=False, overwrite=True)
            # eigenvalues are homogeneous of degree one: the O(n) vector is scaled instead of the n x n Gram
            if scale is not None:
                gram_evals *= scale
            if verbose:
                print(f"Group {group_id}: Store 'gram_evals'")
            evals[group_id] = gram_evals

        return group_hook

    @staticmethod
    def _check_param_groups(param_groups: List[Dict]):
        check_key_exists(param_groups, "params")
        check_unique_params(param_groups)
