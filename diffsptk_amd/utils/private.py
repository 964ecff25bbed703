"""Small host helpers with the reference's behaviour (diffsptk/utils/private.py)."""
from __future__ import annotations

from itertools import islice
from typing import Any, Callable

import numpy as np
import torch


def check_size(x: int, y: int, cause: str) -> None:
    # private.py:97-99
    if x != y:
        raise ValueError(f"Unexpected {cause} (input {x} vs target {y}).")


def filter_values(d: dict[str, Any], drop_keys=()) -> dict[str, Any]:
    # private.py:63-72: strip the implicit names out of locals()
    return {k: v for k, v in d.items() if k not in ("self", "__class__", *drop_keys)}


def get_layer(is_module: bool, module, params: dict[str, Any]) -> Callable:
    """private.py:45-60: a sub-module instance, or a closure over ``module._func``."""
    if is_module:
        return module(**params)
    if module._takes_input_size:
        params = dict(islice(params.items(), 1, None))
    params = {k: v for k, v in params.items() if k not in ("learnable", "device", "dtype")}

    def layer(*args, **kwargs):
        return module._func(*args, **params, **kwargs)

    return layer


def to(x, device=None, dtype=None) -> torch.Tensor:
    """private.py:134-154: float64 numpy/torch table -> tensor of the module dtype."""
    if dtype is None:
        dtype = torch.get_default_dtype()
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=device, dtype=dtype)
