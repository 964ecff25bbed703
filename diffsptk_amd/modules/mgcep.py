"""Mel-generalized cepstral analysis (reference: mgcep.py) -- SURVEY.md section 8(f), row 3.

gamma = 0 is mel-cepstral analysis (the tuned kernel of modules/mcep.py, as in the reference: mgcep.py:97-105).
For gamma in [-1, 0) every linear stage of a Newton step (mgcep.py:185-249: cfreqt + rfft; irfft + pfreqt / rfreqt
+ the P / Q transforms) is composed on the host into float64 matrices (utils/tables.py:mgcep_matrices), so a step is
seven row products on the library's freqt kernel around the pointwise spectrum arithmetic, followed by the
Toeplitz-plus-Hankel solve kernel (csrc/mgc.hip).  Gradients: the kernels' own backward entries chained by autograd.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .. import ops
from ..utils import tables
from ..utils.private import check_size, to
from .gnorm import GeneralizedCepstrumGainNormalization as _Gnorm
from .gnorm import GeneralizedCepstrumInverseGainNormalization as _Ignorm
from .gnorm import get_gamma
from .mc2b import MelCepstrumToMLSADigitalFilterCoefficients, MLSADigitalFilterCoefficientsToMelCepstrum
from .mcep import MelCepstralAnalysis
from .mgc2mgc import MelGeneralizedCepstrumToMelGeneralizedCepstrum

_NAMES = ("Cr", "Ci", "Pr", "Qr", "Qi", "Rr", "Ri", "R1", "Q1")


class MelGeneralizedCepstralAnalysis(nn.Module):
    """x:(..., L/2+1) power spectrum -> mel-generalized cepstrum (..., M+1) (mgcep.py:181-249)."""

    def __init__(self, *, fft_length: int, cep_order: int, alpha: float = 0, gamma: float = 0, c: int | None = None,
                 n_iter: int = 0, device=None, dtype=None) -> None:
        super().__init__()
        gamma = get_gamma(gamma, c)
        if fft_length <= 1:
            raise ValueError("fft_length must be greater than 1.")
        if cep_order < 0:
            raise ValueError("cep_order must be non-negative.")
        if fft_length < 2 * cep_order:
            raise ValueError("cep_order must be less than or equal to fft_length // 2.")
        if 1 <= abs(alpha):
            raise ValueError("alpha must be in (-1, 1).")
        if gamma < -1 or 0 < gamma:
            raise ValueError("gamma must be in [-1, 0].")
        if n_iter < 0:
            raise ValueError("n_iter must be non-negative.")
        self.fft_length, self.cep_order, self.gamma, self.n_iter = fft_length, cep_order, gamma, n_iter
        if gamma == 0:
            self.mcep = MelCepstralAnalysis(fft_length=fft_length, cep_order=cep_order, alpha=alpha, n_iter=n_iter,
                                            device=device, dtype=dtype)
            return
        if cep_order < 1:
            raise ValueError("cep_order must be positive when gamma is not 0.")
        if cep_order > 64:   # the Toeplitz-plus-Hankel solve keeps a system per wave (csrc/mgc.hip): say so here, not at the first call
            raise ValueError("cep_order must be at most 64 when gamma is not 0 (limit of the device solver).")
        M = cep_order
        mats = dict(tables.mgcep_matrices(fft_length, cep_order, float(alpha)))   # (the table function caches its result)
        # only the columns the step uses (mgcep.py:226-227: pt = p[:M], qt = q[2:]): one 48-column launch per product
        mats["Pr"] = mats["Pr"][:, :M]
        for k in ("Qr", "Qi", "Q1"):
            mats[k] = mats[k][:, 2:]
        for name, mat in mats.items():
            self.register_buffer(name, to(np.ascontiguousarray(mat), device=device, dtype=dtype), persistent=False)
        # float32 / fft_length 512 / cep_order <= 24: a step's spectrum arithmetic and its five row products run as ONE launch on
        # operand images in matrix-instruction order (dsa_mgcep_step)
        if fft_length == 512 and cep_order <= 24 and (dtype or torch.get_default_dtype()) == torch.float32:
            self.register_buffer("step_images", to(tables.mgcep_step_images(fft_length, cep_order, float(alpha)), device=device,
                                                   dtype=torch.float32), persistent=False)
            self.register_buffer("step_images_bwd", to(tables.mgcep_step_bwd_images(fft_length, cep_order, float(alpha)),
                                                       device=device, dtype=torch.float32), persistent=False)
        else:
            self.step_images = None
            self.step_images_bwd = None
        # ... and at cep_order 24 the WHOLE step (chains as binary16 splits + the solve + the update) is one launch without a graph
        # (dsa_mgcep_step_solve; DSA_MGCEP_STEP_SOLVE=0: the two launches of rounds 2-4, for A/B runs)
        if self.step_images is not None and cep_order == 24 and -1 < gamma < 0:
            self.register_buffer("step_images_h", torch.from_numpy(tables.mgcep_step_h_buffer(fft_length, cep_order, float(alpha))).to(device),
                                 persistent=False)
            # ... and the step's adjoint on the binary16 matrix pipe too (DSA_MGCEP_STEP_BWD_H=0: the float32 kernel, for A/B runs)
            if os.environ.get("DSA_MGCEP_STEP_BWD_H") != "0":
                # (stored as int16 bit patterns like step_images_h's bytes: nn.Module.float() / .to(dtype) cast every FLOATING buffer, and
                #  a binary16 image silently cast to float32 would be read by the float32 kernel in the wrong layout)
                self.register_buffer("step_images_bwd", torch.from_numpy(tables.mgcep_step_bwd_h_images(fft_length, cep_order, float(alpha))
                                                                         .view("int16")).to(device), persistent=False)
        else:
            self.step_images_h = None
        self._zero_q = None   # (see forward: the Hankel generator of the gamma = -1 step)
        self._zero_q_ready = None
        self.b2mc = MLSADigitalFilterCoefficientsToMelCepstrum(M, alpha, device=device, dtype=dtype)
        self.mc2b = MelCepstrumToMLSADigitalFilterCoefficients(M, alpha, device=device, dtype=dtype)
        self.gc2gc = MelGeneralizedCepstrumToMelGeneralizedCepstrum(M, M, in_gamma=-1, out_gamma=gamma, device=device,
                                                                    dtype=dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.gamma == 0:
            return self.mcep(x)
        M, H = self.cep_order, self.fft_length // 2
        check_size(x.size(-1), H + 1, "dimension of spectrum")
        mm = ops.MatmulRowsFn.apply

        def epsilon(gamma, r, b1):
            return r[..., 0] + gamma * (r[..., 1:] * b1).sum(-1)

        def newton(gamma, b1, need_gain=True):   # need_gain: b0 = sqrt(eps) is only read after the LAST step of a run
            b1_old = b1
            if gamma != -1 and self.step_images_h is not None and x.dtype == torch.float32 and self.step_images_h.device == x.device \
                    and os.environ.get("DSA_MGCEP_STEP_SOLVE") != "0":
                # mgcep.py:199-230 in ONE launch; with a graph wanted the same launch keeps (pt, qt) and the backward is the adjoint
                # solve + the step's adjoint (ops.MgcepStepSolveFn)
                if torch.is_grad_enabled() and (x.requires_grad or b1.requires_grad):
                    b1, r = ops.MgcepStepSolveFn.apply(x, b1, self.step_images_h, self.step_images_bwd, gamma)
                    if not need_gain:
                        return None, b1, None
                    return torch.sqrt(epsilon(gamma, r, b1_old)).unsqueeze(-1), b1, None   # mgcep.py:221 (b_eps = the step's input coefficients)
                b1, r = ops.mgcep_step_solve(x, b1, self.step_images_h, gamma)
                if not need_gain:
                    return None, b1, None
                if fused_gain:
                    return None, b1, ops.mgcep_gain(r, b1_old, gamma, b1)
                return torch.sqrt(epsilon(gamma, r, b1_old)).unsqueeze(-1), b1, None
            if gamma == -1:                                        # mgcep.py:196-197, 213-215
                pt = mm(x, self.Pr)
                qt = None                                          # q (1 + gamma) = 0: no Hankel part
                r = mm(x, self.R1)
            elif not (torch.is_grad_enabled() and (x.requires_grad or b1.requires_grad)) and self.step_images is not None \
                    and x.dtype == torch.float32 and self.step_images.device == x.device:
                pt, qt, r = ops.mgcep_step(x, b1, self.step_images, gamma)   # mgcep.py:199-220 in one launch (forward only)
            elif self.step_images is not None and x.dtype == torch.float32 and self.step_images.device == x.device:
                # a graph is wanted: the same launch forward, its adjoint as one launch backward (ops.MgcepStepFn)
                pt, qt, r = ops.MgcepStepFn.apply(x, b1, self.step_images, self.step_images_bwd, gamma)
            elif not (torch.is_grad_enabled() and (x.requires_grad or b1.requires_grad)) and M <= 64:
                S = ops.mgcep_spectra(x, b1, self.Cr, self.Ci, gamma)   # mgcep.py:199-209, one pass (forward only)
                pt = mm(S[0], self.Pr)
                qt = (mm(S[1], self.Qr) + mm(S[2], self.Qi)) * (1 + gamma)
                r = mm(S[3], self.Rr) + mm(S[4], self.Ri)
            else:
                b = torch.cat((torch.zeros_like(b1[..., :1]), b1), dim=-1)
                X = 1 + gamma * mm(b, self.Cr)                     # mgcep.py:199-209
                Y = gamma * mm(b, self.Ci)
                XX, YY = X * X, Y * Y
                D = XX + YY
                pp = x * torch.pow(D, -1 / gamma) / D
                qq = pp / D
                pt = mm(pp, self.Pr)
                qt = (mm(qq * (XX - YY), self.Qr) + mm(qq * (2 * X * Y), self.Qi)) * (1 + gamma)
                r = mm(pp * X, self.Rr) + mm(pp * Y, self.Ri)
            if qt is None:
                # q (1 + gamma) = 0 at gamma = -1: a block of zeros the solve only reads -- kept from call to call (no fill launch) unless
                # a graph or a stream capture could make it outlive this call's view of it
                shape = (*pt.shape[:-1], 2 * M - 1)
                zq = self._zero_q
                with torch.cuda.device(pt.device):   # events and streams below belong to the INPUT's device, not the current one
                    cur = torch.cuda.current_stream()
                    nbytes = pt.element_size() * (pt.numel() // max(M, 1)) * (2 * M - 1)
                    cacheable = not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing() and nbytes <= (16 << 20)
                    if not cacheable or zq is None or zq.shape != shape or zq.device != pt.device or zq.dtype != pt.dtype:
                        if zq is not None and zq.is_cuda and zq.device == pt.device:
                            zq.record_stream(cur)   # the block being dropped may still be read by launches queued on this stream
                        zq = torch.zeros(shape, device=pt.device, dtype=pt.dtype)
                        if cacheable:               # (blocks above 16 MB are not kept for the module's lifetime)
                            self._zero_q = zq
                            self._zero_q_ready = torch.cuda.Event()
                            self._zero_q_ready.record(cur)
                    else:
                        if not self._zero_q_ready.query():   # filled on another stream and possibly not done yet
                            cur.wait_event(self._zero_q_ready)
                        zq.record_stream(cur)                # read on this stream: the allocator must not recycle it under us
                qt = zq
            upd = None
            if not (torch.is_grad_enabled() and (pt.requires_grad or qt.requires_grad or r.requires_grad or b1.requires_grad)):
                upd = ops.thsolve_update(pt, qt, r, b1)            # solve + update in one call, r read in place
            b1 = upd if upd is not None else b1 + ops.ThSolveFn.apply(pt, qt, r[..., 1:])      # mgcep.py:226-230
            if not need_gain:
                return None, b1, None
            b_eps = b1 if gamma == -1 else b1_old     # mgcep.py:213-215 / 221: the gain uses the updated coefficients only at gamma = -1
            if fused_gain and not (torch.is_grad_enabled() and (r.requires_grad or b1.requires_grad or b_eps.requires_grad)):
                return None, b1, ops.mgcep_gain(r, b_eps, gamma, b1)   # (b0, b1) in one launch, already joined
            return torch.sqrt(epsilon(gamma, r, b_eps)).unsqueeze(-1), b1, None

        # without a gradient the gain and the joining of (b0, b1) are one launch (dsa_mgcep_gain) instead of six stock ones
        fused_gain = x.is_cuda and x.dtype in (torch.float32, torch.float64) and M >= 1   # (dsa_mgcep_gain wants cep_order >= 1)
        b1 = torch.zeros(*x.shape[:-1], M, device=x.device, dtype=x.dtype)
        b0, b1, b = newton(-1, b1)
        if self.gamma != -1:
            if b is None:
                b = torch.cat((b0, b1), dim=-1)
            if b0 is None:
                b0 = b[..., :1]        # mgcep.py:240-249: with n_iter = 0 the gain of the gamma = -1 step is what is returned
            b = _Gnorm._forward(self.mc2b(self.gc2gc(self.b2mc(_Ignorm._forward(b, gamma=-1)))), gamma=self.gamma)   # b2b, :120-137
            b1 = b[..., 1:]            # mgcep.py:244: only b1 of b2b's output is kept
            b = None
            one_launch = self.n_iter >= 1 and self.step_images_h is not None and x.dtype == torch.float32 and self.step_images_h.device == x.device \
                and not (torch.is_grad_enabled() and (x.requires_grad or b1.requires_grad)) and os.environ.get("DSA_MGCEP_STEP_SOLVE") != "0" \
                and os.environ.get("DSA_MGCEP_ALL_STEPS") != "0"
            if one_launch:
                # no graph wanted: ALL n_iter Newton steps in one launch (a frame's iteration depends on the frame alone), then the gain
                b1, r, b1_prev = ops.mgcep_step_solve(x, b1, self.step_images_h, self.gamma, n_steps=self.n_iter, want_prev=True)
                if fused_gain:
                    b = ops.mgcep_gain(r, b1_prev, self.gamma, b1)
                else:
                    b0 = torch.sqrt(epsilon(self.gamma, r, b1_prev)).unsqueeze(-1)
            for it in range(0 if one_launch else self.n_iter):
                last = it == self.n_iter - 1
                b0_it, b1, b_it = newton(self.gamma, b1, need_gain=last)
                if last:
                    b0, b = b0_it, b_it
        if b is None:
            b = torch.cat((b0, b1), dim=-1)
        return self.b2mc(_Ignorm._forward(b, gamma=self.gamma))                                                       # b2mc, :139-144
