"""Mel-cepstral analysis (reference: diffsptk/modules/mcep.py)."""
from __future__ import annotations

import torch

from .. import _lib, ops
from ..utils import tables
from ..utils.private import check_size, filter_values, to
from .base import BaseFunctionalModule, Precomputed


class MelCepstralAnalysis(BaseFunctionalModule):
    """Power spectrum (..., L/2+1) -> mel-cepstrum (..., M+1) by ``n_iter`` Newton steps on the
    Toeplitz-plus-Hankel system (mcep.py:189-224).

    The reference's three warping matrices and the FFTs between them are composed on the host
    into G, D, E (utils/tables.py:mcep_matrices); the whole iteration then runs in one kernel
    with only X and the result touching HBM.
    """

    _takes_input_size = True

    def __init__(self, *, fft_length: int, cep_order: int, alpha: float = 0, n_iter: int = 0,
                 device=None, dtype=None) -> None:
        super().__init__()
        self.in_dim = fft_length // 2 + 1
        self._register_precomputed(self._precompute(**filter_values(locals())))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        check_size(x.size(-1), self.in_dim, "dimension of spectrum")
        return self._call_forward(x)

    @staticmethod
    def _func(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        pre = MelCepstralAnalysis._precompute(2 * x.size(-1) - 2, *args, **kwargs, dtype=x.dtype,
                                              device=x.device, module=False)
        return MelCepstralAnalysis._apply_precomputed(pre, x=x)

    @staticmethod
    def _check(fft_length: int, cep_order: int, alpha: float, n_iter: int) -> None:
        if fft_length <= 1:
            raise ValueError("fft_length must be greater than 1.")
        if cep_order < 0:
            raise ValueError("cep_order must be non-negative.")
        if fft_length < 2 * cep_order:
            raise ValueError("cep_order must be less than or equal to fft_length // 2.")
        if 1 <= abs(alpha):
            raise ValueError("alpha must be in (-1, 1).")
        if n_iter < 0:
            raise ValueError("n_iter must be non-negative.")

    @staticmethod
    def _precompute(fft_length: int, cep_order: int, alpha: float, n_iter: int, device, dtype,
                    module: bool = True) -> Precomputed:
        MelCepstralAnalysis._check(fft_length, cep_order, alpha, n_iter)
        tens = _device_matrices(fft_length, cep_order, float(alpha), device, dtype, cache=not module)
        # The tuned kernels hold the warping matrices as scaled binary16 hi/lo images whose scales are
        # chosen for |alpha| <= 0.95 (csrc/mcep_mfma_f16.h); more extreme warping stays on the generic kernel.
        algo = _lib.ALGO_AUTO if abs(alpha) <= 0.95 else _lib.ALGO_GENERIC
        return Precomputed(values={"fft_length": fft_length, "cep_order": cep_order, "n_iter": n_iter, "algo": algo},
                           tensors=tens)

    @staticmethod
    def _forward(x: torch.Tensor, *, fft_length: int, cep_order: int, n_iter: int, G: torch.Tensor,
                 D: torch.Tensor, E: torch.Tensor, alpha_vector: torch.Tensor, algo: int = _lib.ALGO_AUTO) -> torch.Tensor:
        if x.requires_grad and torch.is_grad_enabled():
            # no tuned kernel for this geometry and a graph is wanted: the differentiable whole-batch composition
            y = ops.mcep_composed(x, G, D, E, alpha_vector, fft_length, cep_order, n_iter, algo)
            if y is not None:
                return y
        return ops.McepFn.apply(x, G, D, E, alpha_vector, fft_length, cep_order, n_iter, algo)


_MAT_CACHE: dict = {}


def _device_matrices(fft_length, cep_order, alpha, device, dtype, cache):
    """G, D, E, alpha_vector on the target device.  The functional path caches them (the
    reference recomputes its matrices -- a 257x49 Python double loop -- on every call)."""
    key = (fft_length, cep_order, alpha, str(device), dtype)
    if cache and key in _MAT_CACHE:
        return _MAT_CACHE[key]
    G, D, E, av = tables.mcep_matrices(fft_length, cep_order, alpha)[:4]
    tens = {"G": to(G, device, dtype), "D": to(D, device, dtype), "E": to(E, device, dtype),
            "alpha_vector": to(av, device, dtype)}
    if tens["G"].device.type == "cuda":   # the tuned kernels' operand images: prepared here, once (ops.mcep_images)
        ops.mcep_images(tens["G"], tens["D"], tens["E"], fft_length, cep_order)
    if cache:
        _MAT_CACHE[key] = tens
    return tens
