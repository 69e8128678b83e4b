"""Extension-hook scheduling for BackPACK-style backward passes.

Behavioural mirror of ``vivit/utils/hooks.py`` (the host-side boundary of the path; no kernels
here): a hook object is called once per module right after the extensions ran on it; it visits
each trainable parameter exactly once, skips ``Sequential`` containers, stores per-parameter
results under ``savefield``, and -- for the group variant -- folds per-parameter results into a
per-group accumulator and fires the group computation the moment the last parameter of a group
has been back-propagated.  Hook objects are single-use (``processed`` is never reset).
"""
from __future__ import annotations

import types
from typing import Any, Callable, Dict, List

from torch.nn import Module, Parameter, Sequential


class ModuleHook:
    """Visit parameters with access to their module.  Subclasses implement ``module_hook``."""

    def __init__(self, savefield: str = None):
        self.savefield = savefield  # None: side effects only, nothing is attached to parameters
        self.processed = set()

    def module_hook(self, param: Parameter, module: Module) -> Any:
        raise NotImplementedError

    def __call__(self, module: Module):
        for param in module.parameters():
            if self.should_run_hook(param, module):
                self.run_hook(param, module)

    def should_run_hook(self, param: Parameter, module: Module) -> bool:
        # containers see their children's parameters again: only leaf-like modules count
        # (vivit/utils/hooks.py:57-73)
        if isinstance(module, Sequential):
            return False
        return param.requires_grad and id(param) not in self.processed

    def run_hook(self, param: Parameter, module: Module):
        self._save(self.module_hook(param, module), param)
        self.processed.add(id(param))

    def _save(self, value: Any, param: Parameter):
        if self.savefield is None:
            if value is not None:
                raise ValueError(f"Hook has no savefield, but produced output of type {type(value)}.")
            return
        setattr(param, self.savefield, value)


class ParameterHook(ModuleHook):
    """Visit parameters without module access.  Subclasses implement ``param_hook``."""

    def param_hook(self, param: Parameter) -> Any:
        raise NotImplementedError

    def module_hook(self, param: Parameter, module: Module) -> Any:
        return self.param_hook(param)


class ParameterGroupsHook(ParameterHook):
    """Per-parameter work, accumulated per group, finalised when a group is complete.

    Subclass (or use :meth:`from_functions`) to provide ``param_computation(param)``,
    ``accumulate(existing, update)`` and ``group_hook(accumulation, group)``.
    """

    def __init__(self, param_groups: List[Dict[str, Any]]):
        super().__init__(None)
        flat = [id(p) for g in param_groups for p in g["params"]]
        if len(flat) != len(set(flat)):
            raise ValueError("Same parameters occur in different groups")
        self._param_groups = param_groups
        self._groups_by_id = {id(g): g for g in param_groups}
        self._param_to_group = {id(p): id(g) for g in param_groups for p in g["params"]}
        self._pending = {id(g): {id(p) for p in g["params"]} for g in param_groups}
        self._accumulations: Dict[int, Any] = {}
        self._output: Dict[int, Any] = {}
        self._processed_groups = set()

    # -- to be provided -------------------------------------------------------------------
    def param_computation(self, param: Parameter) -> Any:
        raise NotImplementedError

    def accumulate(self, existing: Any, update: Any) -> Any:
        raise NotImplementedError

    def group_hook(self, accumulation: Any, group: Dict[str, Any]) -> Any:
        raise NotImplementedError

    # -- scheduling -----------------------------------------------------------------------
    def should_run_hook(self, param: Parameter, module: Module) -> bool:
        return id(param) in self._param_to_group and super().should_run_hook(param, module)

    def current_accumulation(self, param: Parameter) -> Any:
        """The group accumulator ``param`` will be folded into (``None`` before the first param).

        Lets a ``param_computation`` accumulate in place on the device (beta = 1 in the Gram
        kernel) instead of materialising a temporary per parameter.
        """
        return self._accumulations.get(self._param_to_group[id(param)])

    def should_run_group_hook(self, param: Parameter) -> bool:
        pending = self._pending[self._param_to_group[id(param)]]
        return id(param) not in self.processed and pending == {id(param)}

    def param_hook(self, param: Parameter):
        gid = self._param_to_group[id(param)]
        result = self.param_computation(param)
        if gid in self._accumulations:
            self._accumulations[gid] = self.accumulate(self._accumulations[gid], result)
        else:
            self._accumulations[gid] = result
        last = self.should_run_group_hook(param)
        self._pending[gid].discard(id(param))
        if last:
            self.run_group_hook(gid)

    def run_group_hook(self, group_id: int):
        accumulation = self._accumulations.pop(group_id)
        self._output[group_id] = self.group_hook(accumulation, self.get_group(group_id))
        self._processed_groups.add(group_id)

    def get_group(self, group_id: int) -> Dict[str, Any]:
        return self._groups_by_id[group_id]

    def get_output(self, group: Dict[str, Any], pop: bool = True) -> Any:
        """Result of ``group_hook`` for ``group``; ValueError while the group is incomplete."""
        if any(id(p) not in self.processed for p in group["params"]):
            raise ValueError("Group contains unprocessed parameters.")
        return self._output.pop(id(group)) if pop else self._output[id(group)]

    @classmethod
    def from_functions(
        cls,
        param_groups: List[Dict[str, Any]],
        param_computation_fn: Callable[[ParameterGroupsHook, Parameter], Any],
        group_hook_fn: Callable[[ParameterGroupsHook, Any, Dict[str, Any]], Any],
        accumulate_fn: Callable[[ParameterGroupsHook, Any, Any], Any],
    ) -> ParameterGroupsHook:
        """Build a hook from three plain functions taking the hook as first argument."""
        hook = cls(param_groups)
        hook.param_computation = types.MethodType(param_computation_fn, hook)
        hook.group_hook = types.MethodType(group_hook_fn, hook)
        hook.accumulate = types.MethodType(accumulate_fn, hook)
        return hook
