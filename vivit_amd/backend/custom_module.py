"""Branching / padding / slicing / scaling modules of the stand-in backend.

Counterparts of ``backpack.custom_module.{branching,pad,slicing,scale_module}`` which the reference's ViViT extensions
support (vivit/extensions/secondorder/vivit/__init__.py:113-117) and its tests use for skip connections
(test/settings.py:161-181): residual networks are written as ``Parallel(branch_1, ..., branch_k)`` whose outputs a
``SumModule`` adds, so that every tensor operation belongs to a leaf module the extensions can differentiate.
"""
from typing import Sequence, Tuple, Union

import torch
from torch import Tensor, nn
from torch.nn import functional as F


class SumModule(nn.Module):
    """Sum of its inputs (``backpack.custom_module.branching.SumModule``)."""

    def forward(self, *inputs: Tensor) -> Tensor:
        out = inputs[0]
        for t in inputs[1:]:
            out = out + t
        return out if len(inputs) > 1 else out * 1.0


class ActiveIdentity(nn.Module):
    """Identity that produces a new tensor (``backpack.custom_module.branching.ActiveIdentity``)."""

    def forward(self, x: Tensor) -> Tensor:
        return x * 1.0


class Parallel(nn.Module):
    """Feed the input to every branch, merge the results (``backpack.custom_module.branching.Parallel``)."""

    def __init__(self, *branches: nn.Module, merge_module: nn.Module = None):
        super().__init__()
        for i, b in enumerate(branches):
            self.add_module(f"branch{i}", b)
        self._num = len(branches)
        self.merge = SumModule() if merge_module is None else merge_module

    def forward(self, x: Tensor) -> Tensor:
        return self.merge(*[getattr(self, f"branch{i}")(x) for i in range(self._num)])


class Pad(nn.Module):
    """``torch.nn.functional.pad`` as a module (``backpack.custom_module.pad.Pad``)."""

    def __init__(self, pad: Sequence[int], mode: str = "constant", value: float = 0.0):
        super().__init__()
        self.pad, self.mode, self.value = tuple(pad), mode, value

    def forward(self, x: Tensor) -> Tensor:
        return F.pad(x, self.pad, mode=self.mode, value=self.value)


class Slicing(nn.Module):
    """``x[slice_info]`` as a module (``backpack.custom_module.slicing.Slicing``)."""

    def __init__(self, slice_info: Tuple[Union[slice, int], ...]):
        super().__init__()
        self.slice_info = slice_info

    def forward(self, x: Tensor) -> Tensor:
        return x[self.slice_info]


class ScaleModule(nn.Module):
    """``x * weight`` with a constant scalar (``backpack.custom_module.scale_module.ScaleModule``)."""

    def __init__(self, weight: float = 1.0):
        super().__init__()
        self.weight = float(weight)

    def forward(self, x: Tensor) -> Tensor:
        return x * self.weight
