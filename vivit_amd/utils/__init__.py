"""Utilities shared by the Computation classes (mirror of ``vivit.utils``)."""
from torch import Tensor


def delete_savefield(param: Tensor, savefield: str, verbose: bool = False):
    """Drop ``param.<savefield>`` as soon as it has been consumed (vivit/utils/__init__.py:8-19)."""
    if verbose:
        print(f"Param {id(param)}: Delete '{savefield}'")
    delattr(param, savefield)
