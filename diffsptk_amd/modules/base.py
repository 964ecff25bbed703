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


def _reference_keys_on_save(module, state, prefix, _meta):
    for local, (ref_key, ref_shape) in module._reference_state_keys.items():
        k = prefix + local
        if k in state:
            t = state.pop(k)
            state[prefix + ref_key] = t.reshape(ref_shape) if ref_shape is not None else t


def _reference_keys_on_load(module, state, prefix, *_rest):
    for local, (ref_key, _ref_shape) in module._reference_state_keys.items():
        k = prefix + ref_key
        if k in state:
            t = state.pop(k)
            own = module._parameters.get(local)
            state[prefix + local] = t.reshape(own.shape) if own is not None and t.numel() == own.numel() else t


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
        self._install_reference_state_keys()

    #: ``state_dict`` compatibility with the reference for COMPOSITE modules, whose learnable tensors live in sub-modules there
    #: (e.g. ``window.window`` / ``spec.fftr.W`` of diffsptk.STFT, stft.py:186-235) and directly on the module here (one fused
    #: launch, no sub-modules): {local name: (reference key, reference shape or None)}.  ``state_dict()`` writes the reference's
    #: keys and shapes, ``load_state_dict()`` reads them (and still accepts the local names): checkpoints interchange.
    _reference_state_keys: ClassVar[dict[str, tuple[str, tuple[int, ...] | None]]] = {}

    def _install_reference_state_keys(self) -> None:
        if self._reference_state_keys:
            # (module-level functions, not closures: a module with hooks must stay picklable -- torch.save(model))
            self._register_state_dict_hook(_reference_keys_on_save)
            self._register_load_state_dict_pre_hook(_reference_keys_on_load, with_module=True)

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
