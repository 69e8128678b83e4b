"""Parameter-group validation; same error conventions as vivit/utils/checks.py:6-49."""
from typing import Dict, List, Optional


def check_key_exists(param_groups: List[Dict], key: str):
    """Every group must define ``key`` (ValueError otherwise)."""
    for group in param_groups:
        if key not in group:
            raise ValueError(f"At least one group is not specifying '{key}'.")


def check_unique_params(param_groups: List[Dict]):
    """A parameter may belong to one group only (ValueError otherwise)."""
    seen = set()
    for group in param_groups:
        for p in group["params"]:
            if id(p) in seen:
                raise ValueError("At least one parameter is in more than one group.")
            seen.add(id(p))


def check_subsampling_unique(subsampling: Optional[List[int]]):
    """Sub-sampling indices must not repeat (ValueError otherwise); ``None`` = full batch."""
    if subsampling is None:
        return
    if len(set(subsampling)) != len(subsampling):
        raise ValueError("Detected repeated index in subsampling.")
