"""Operator contract shared by every module (mirrors diffsptk/modules/base.py:26-101).

A functional module exposes, besides ``forward``:
  ``_check``      validate constructor options (raises ValueError, same messages as the reference)
  ``_precompute`` options -> ``Precomputed(values, layers, tensors)``
  ``_forward``    static, pure: ``_forward(x, **state)`` with the state bound BY NAME
  ``_func``       functional entry: precompute on the fly for ``x.device / x.dtype``
so that ``diffsptk_amd.functional.f(x, ...)`` and ``Module(...)(x)`` run the same code.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Callable, ClassVar, NamedTuple

import torch
from torch import nn


class Precomputed(NamedTuple):
    values: dict[str, Any] = {}
    layers: dict[str, Callable] = {}
    tensors: dict[str, torch.Tensor] = {}


class BaseFunctionalModule(ABC, nn.Module):
    #: True when the first parameter of ``_precompute`` is inferred from the input on the
    #: functional path (base.py:42, utils/private.py:51-52 of the reference)
    _takes_input_size: ClassVar[bool] = False

    _value_names: tuple[str, ...] = ()
    _layer_names: tuple[str, ...] = ()

    def _register_precomputed(self, pre: Precomputed, learnable=False) -> None:
        """values/layers become attributes, tensors become non-persistent buffers -- or
        Parameters when learnable (a bool, or a collection naming the learnable tensors)."""
        self._value_names = tuple(pre.values)
        self._layer_names = tuple(pre.layers)
        for k, v in {**pre.values, **pre.layers}.items():
            setattr(self, k, v)
        for k, t in pre.tensors.items():
            is_param = learnable is True or (not isinstance(learnable, bool) and k in learnable)
            if is_param:
                setattr(self, k, nn.Parameter(t))
            else:
                self.register_buffer(k, t, persistent=False)  # state_dict stays empty (base.py:67)

    def _state(self) -> dict[str, Any]:
        st = {k: getattr(self, k) for k in self._value_names + self._layer_names}
        st.update(self._buffers)
        st.update(self._parameters)
        return st

    def _call_forward(self, *inputs):
        return self._forward(*inputs, **self._state())

    @classmethod
    def _apply_precomputed(cls, pre: Precomputed, **inputs):
        return cls._forward(**inputs, **pre.values, **pre.layers, **pre.tensors)

    @staticmethod
    @abstractmethod
    def _func(*args, **kwargs):
        ...

    @staticmethod
    @abstractmethod
    def _check(*args, **kwargs) -> None:
        ...

    @staticmethod
    @abstractmethod
    def _precompute(*args, **kwargs) -> Precomputed:
        ...

    @staticmethod
    @abstractmethod
    def _forward(*args, **kwargs):
        ...
