"""torch.autograd.Function wrappers over the C-ABI (include/diffsptk_amd.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation below is a
call into libdiffsptk_amd.so.  Inputs must live on a HIP device -- there is deliberately no CPU
path (the reference's autograd-derived backward, SURVEY.md section 3.5, is replaced by the
hand-written backward kernels).
"""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _lib

_PAD = {"constant": 0, "reflect": 1, "replicate": 2, "circular": 3}


def pad_mode_code(mode: str) -> int:
    try:
        return _PAD[mode]
    except KeyError:
        raise ValueError(f"mode {mode} is not supported.") from None


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.float64:
        return _lib.F64
    raise TypeError(f"diffsptk_amd supports float32/float64 tensors, got {t.dtype}")


def _require_device(*tensors) -> None:
    for t in tensors:
        if t is not None and t.device.type != "cuda":
            raise RuntimeError(
                "diffsptk_amd is a HIP (MI355X) device backend: expected tensors on a 'cuda' "
                f"(ROCm) device, got {t.device}.  There is no CPU fallback."
            )


def _same_dtype(ref: torch.Tensor, *others) -> None:
    for t in others:
        if t is not None and t.dtype != ref.dtype:
            raise RuntimeError(f"expected scalar type {ref.dtype} but found {t.dtype}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args), name)


def num_frames(T: int, P: int) -> int:
    return 0 if T <= 0 else (T - 1) // P + 1


# ----------------------------------------------------------------------------------- Frame
class FrameFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, L, P, center, zmean, mode):
        _require_device(x)
        xc = x.contiguous()
        T = xc.size(-1)
        B = xc.numel() // T if T > 0 else 0
        N = num_frames(T, P)
        y = torch.empty(*xc.shape[:-1], N, L, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _call("dsa_frame_fwd", _p(xc), B, T, L, P, int(center), int(zmean), pad_mode_code(mode),
                  _dtype_code(xc), _p(y), _stream())
        ctx.cfg = (xc.shape, L, P, center, zmean, mode)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        shape, L, P, center, zmean, mode = ctx.cfg
        gy = gy.contiguous()
        T = shape[-1]
        B = gy.numel() // (num_frames(T, P) * L)
        gx = torch.empty(shape, device=gy.device, dtype=gy.dtype)
        with torch.cuda.device(gy.device):
            _call("dsa_frame_bwd", _p(gy), B, T, L, P, int(center), int(zmean), pad_mode_code(mode),
                  _dtype_code(gy), _p(gx), _stream())
        return gx, None, None, None, None, None


# ----------------------------------------------------------------------------------- Window
class WindowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, out_length):
        _require_device(x, w)
        _same_dtype(x, w)
        xc, wc = x.contiguous(), w.contiguous()
        L = xc.size(-1)
        L2 = L if out_length is None else out_length
        F = xc.numel() // L
        y = torch.empty(*xc.shape[:-1], L2, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _call("dsa_window_fwd", _p(xc), F, L, _p(wc), L2, _dtype_code(xc), _p(y), _stream())
        ctx.save_for_backward(xc, wc)
        ctx.L2 = L2
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, wc = ctx.saved_tensors
        gy = gy.contiguous()
        L = xc.size(-1)
        F = xc.numel() // L
        gx = torch.empty_like(xc)
        gw = torch.empty_like(wc) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(gy.device):
            _call("dsa_window_bwd", _p(gy), _p(xc), F, L, _p(wc), ctx.L2, _dtype_code(xc), _p(gx), _p(gw),
                  _stream())
        return gx, gw, None


# ----------------------------------------------------------------------------------- fftr
class FftrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fft_length, fmt, twiddle):
        _require_device(x, twiddle)
        _same_dtype(x, twiddle)
        xc = x.contiguous()
        len_in = xc.size(-1)
        F = xc.numel() // len_in
        K = fft_length // 2 + 1
        shape = (*xc.shape[:-1], K, 2) if fmt == 0 else (*xc.shape[:-1], K)
        y = torch.empty(shape, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _call("dsa_fftr_fwd", _p(xc), F, len_in, fft_length, fmt, _p(twiddle), _dtype_code(xc), _p(y),
                  _stream())
        ctx.save_for_backward(xc, twiddle)
        ctx.cfg = (fft_length, fmt)
        return torch.view_as_complex(y) if fmt == 0 else y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, twiddle = ctx.saved_tensors
        fft_length, fmt = ctx.cfg
        if fmt == 0:
            gy = torch.view_as_real(gy.resolve_conj())
        gy = gy.contiguous()
        len_in = xc.size(-1)
        F = xc.numel() // len_in
        gx = torch.empty_like(xc)
        with torch.cuda.device(gy.device):
            _call("dsa_fftr_bwd", _p(gy), _p(xc), F, len_in, fft_length, fmt, _p(twiddle), _dtype_code(xc),
                  _p(gx), _stream())
        return gx, None, None, None


# ----------------------------------------------------------------------------------- Spectrum
class SpecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, a, fft_length, eps, relative_floor_db, fmt, twiddle):
        ref = b if b is not None else a
        _require_device(b, a, twiddle)
        _same_dtype(ref, b, a, twiddle)
        bc = b.contiguous() if b is not None else None
        ac = a.contiguous() if a is not None else None
        lb = bc.size(-1) if bc is not None else 0
        la = ac.size(-1) if ac is not None else 0
        F = ref.numel() // ref.size(-1)
        K = fft_length // 2 + 1
        y = torch.empty(*ref.shape[:-1], K, device=ref.device, dtype=ref.dtype)
        use_floor = relative_floor_db is not None
        with torch.cuda.device(ref.device):
            _call("dsa_spec_fwd", _p(bc), lb, _p(ac), la, F, fft_length, float(eps), int(use_floor),
                  float(relative_floor_db or 0.0), fmt, _p(twiddle), _dtype_code(ref), _p(y), _stream())
        ctx.save_for_backward(bc, ac, twiddle)
        ctx.cfg = (fft_length, eps, relative_floor_db, fmt)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        bc, ac, twiddle = ctx.saved_tensors
        fft_length, eps, relative_floor_db, fmt = ctx.cfg
        gy = gy.contiguous()
        ref = bc if bc is not None else ac
        F = ref.numel() // ref.size(-1)
        gb = torch.empty_like(bc) if bc is not None else None
        ga = torch.empty_like(ac) if ac is not None else None
        use_floor = relative_floor_db is not None
        with torch.cuda.device(gy.device):
            _call("dsa_spec_bwd", _p(gy), _p(bc), bc.size(-1) if bc is not None else 0, _p(ac),
                  ac.size(-1) if ac is not None else 0, F, fft_length, float(eps), int(use_floor),
                  float(relative_floor_db or 0.0), fmt, _p(twiddle), _dtype_code(ref), _p(gb), _p(ga), _stream())
        return gb, ga, None, None, None, None, None


# ----------------------------------------------------------------------------------- STFT
class StftFn(torch.autograd.Function):
    """Fused Frame + Window + rFFT + Spectrum formatter (stft.py:237-241)."""

    @staticmethod
    def forward(ctx, x, window, twiddle, L, P, fft_length, center, zmean, mode, eps, relative_floor_db, fmt,
                algo):
        _require_device(x, window, twiddle)
        _same_dtype(x, window, twiddle)
        xc, wc = x.contiguous(), window.contiguous()
        T = xc.size(-1)
        B = xc.numel() // T if T > 0 else 0
        N = num_frames(T, P)
        K = fft_length // 2 + 1
        shape = (*xc.shape[:-1], N, K, 2) if fmt == 4 else (*xc.shape[:-1], N, K)
        y = torch.empty(shape, device=x.device, dtype=x.dtype)
        use_floor = relative_floor_db is not None
        with torch.cuda.device(x.device):
            _call("dsa_stft_fwd", _p(xc), B, T, L, P, fft_length, _p(wc), _p(twiddle), int(center), int(zmean),
                  pad_mode_code(mode), float(eps), int(use_floor), float(relative_floor_db or 0.0), fmt,
                  _dtype_code(xc), algo, _p(y), _stream())
        ctx.save_for_backward(xc, wc, twiddle)
        ctx.cfg = (L, P, fft_length, center, zmean, mode, eps, relative_floor_db, fmt, algo)
        return torch.view_as_complex(y) if fmt == 4 else y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, wc, twiddle = ctx.saved_tensors
        L, P, fft_length, center, zmean, mode, eps, relative_floor_db, fmt, algo = ctx.cfg
        if fmt == 4:
            gy = torch.view_as_real(gy.resolve_conj())
        gy = gy.contiguous()
        T = xc.size(-1)
        B = xc.numel() // T
        gx = torch.empty_like(xc)
        gw = torch.empty_like(wc) if ctx.needs_input_grad[1] else None
        use_floor = relative_floor_db is not None
        with torch.cuda.device(gy.device):
            _call("dsa_stft_bwd", _p(gy), _p(xc), B, T, L, P, fft_length, _p(wc), _p(twiddle), int(center),
                  int(zmean), pad_mode_code(mode), float(eps), int(use_floor), float(relative_floor_db or 0.0),
                  fmt, _dtype_code(xc), algo, _p(gx), _p(gw), _stream())
        return (gx, gw) + (None,) * 11


# ----------------------------------------------------------------------------------- freqt
class MatmulRowsFn(torch.autograd.Function):
    """out = c @ A for a fixed (non-learnable) matrix A (freqt.py:141-143, mcep.py:286-288)."""

    @staticmethod
    def forward(ctx, c, A):
        _require_device(c, A)
        _same_dtype(c, A)
        cc, Ac = c.contiguous(), A.contiguous()
        L1, L2 = Ac.shape
        F = cc.numel() // L1
        out = torch.empty(*cc.shape[:-1], L2, device=c.device, dtype=c.dtype)
        with torch.cuda.device(c.device):
            _call("dsa_freqt_fwd", _p(cc), F, L1, _p(Ac), L2, _dtype_code(cc), _p(out), _stream())
        ctx.save_for_backward(Ac)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (Ac,) = ctx.saved_tensors
        g = g.contiguous()
        L1, L2 = Ac.shape
        F = g.numel() // L2
        gc = torch.empty(*g.shape[:-1], L1, device=g.device, dtype=g.dtype)
        with torch.cuda.device(g.device):
            _call("dsa_freqt_bwd", _p(g), F, L1, _p(Ac), L2, _dtype_code(g), _p(gc), _stream())
        return gc, None


# ----------------------------------------------------------------------------------- fbank
class FbankFn(torch.autograd.Function):
    """y, E = mel filter bank outputs and log energy of power spectra (fbank.py:306-321).
    H is a fixed matrix here (a learnable basis is not supported by the kernels)."""

    @staticmethod
    def forward(ctx, x, H, floor, gamma, use_power):
        _require_device(x, H)
        _same_dtype(x, H)
        xc, Hc = x.contiguous(), H.contiguous()
        K, Cn = Hc.shape
        F = xc.numel() // K
        y = torch.empty(*xc.shape[:-1], Cn, device=x.device, dtype=x.dtype)
        E = torch.empty(*xc.shape[:-1], 1, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _call("dsa_fbank_fwd", _p(xc), F, K, _p(Hc), Cn, float(floor), float(gamma), int(bool(use_power)),
                  _dtype_code(xc), _p(y), _p(E), _stream())
        ctx.save_for_backward(xc, Hc)
        ctx.cfg = (float(floor), float(gamma), int(bool(use_power)))
        return y, E

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gE):
        xc, Hc = ctx.saved_tensors
        floor, gamma, use_power = ctx.cfg
        K, Cn = Hc.shape
        F = xc.numel() // K
        gyc = gy.contiguous() if gy is not None else torch.zeros(*xc.shape[:-1], Cn, device=xc.device, dtype=xc.dtype)
        gEc = gE.contiguous() if gE is not None else None
        gx = torch.empty_like(xc)
        with torch.cuda.device(xc.device):
            _call("dsa_fbank_bwd", _p(gyc), _p(gEc) if gEc is not None else None, _p(xc), F, K, _p(Hc), Cn, floor, gamma,
                  use_power, _dtype_code(xc), _p(gx), _stream())
        return gx, None, None, None, None


# ----------------------------------------------------------------------------------- mcep
class McepFn(torch.autograd.Function):
    """MelCepstralAnalysis._forward (mcep.py:189-224) with composed linear stages."""

    @staticmethod
    def forward(ctx, X, G, D, E, av, fft_length, M, n_iter, algo):
        _require_device(X, G, D, E, av)
        _same_dtype(X, G, D, E, av)
        Xc = X.contiguous()
        K = fft_length // 2 + 1
        F = Xc.numel() // K
        mc = torch.empty(*Xc.shape[:-1], M + 1, device=X.device, dtype=X.dtype)
        need_hist = ctx.needs_input_grad[0]
        hist = torch.empty(n_iter + 1, F, M + 1, device=X.device, dtype=X.dtype) if need_hist else None
        with torch.cuda.device(X.device):
            _call("dsa_mcep_fwd", _p(Xc), F, fft_length, M, n_iter, _p(G), _p(D), _p(E), _p(av),
                  _dtype_code(Xc), algo, _p(mc), _p(hist), _stream())
        if need_hist:
            ctx.save_for_backward(Xc, hist, G, D, E, av)
        ctx.cfg = (fft_length, M, n_iter, algo)
        return mc

    @staticmethod
    @once_differentiable
    def backward(ctx, gmc):
        Xc, hist, G, D, E, av = ctx.saved_tensors
        fft_length, M, n_iter, algo = ctx.cfg
        gmc = gmc.contiguous()
        K = fft_length // 2 + 1
        F = Xc.numel() // K
        gX = torch.empty_like(Xc)
        with torch.cuda.device(gmc.device):
            _call("dsa_mcep_bwd", _p(gmc), _p(Xc), _p(hist), F, fft_length, M, n_iter, _p(G), _p(D), _p(E),
                  _p(av), _dtype_code(Xc), algo, _p(gX), _stream())
        return (gX,) + (None,) * 8


# ----------------------------------------------------------------------------------- LPC branch
class AcorrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, M, fmt):
        _require_device(x)
        xc = x.contiguous()
        L = xc.size(-1)
        F = xc.numel() // L
        r = torch.empty(*xc.shape[:-1], M + 1, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _call("dsa_acorr_fwd", _p(xc), F, L, M, fmt, _dtype_code(xc), _p(r), _stream())
        ctx.save_for_backward(xc)
        ctx.cfg = (M, fmt)
        return r

    @staticmethod
    @once_differentiable
    def backward(ctx, gr):
        (xc,) = ctx.saved_tensors
        M, fmt = ctx.cfg
        gr = gr.contiguous()
        L = xc.size(-1)
        F = xc.numel() // L
        gx = torch.empty_like(xc)
        with torch.cuda.device(gr.device):
            _call("dsa_acorr_bwd", _p(gr), _p(xc), F, L, M, fmt, _dtype_code(xc), _p(gx), _stream())
        return gx, None, None


class LevdurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, eps):
        _require_device(r)
        rc = r.contiguous()
        M = rc.size(-1) - 1
        F = rc.numel() // (M + 1)
        out = torch.empty_like(rc)
        with torch.cuda.device(r.device):
            _call("dsa_levdur_fwd", _p(rc), F, M, float(eps), _dtype_code(rc), _p(out), _stream())
        ctx.save_for_backward(rc, out)
        ctx.eps = eps
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        rc, out = ctx.saved_tensors
        g = g.contiguous()
        M = rc.size(-1) - 1
        F = rc.numel() // (M + 1)
        gr = torch.empty_like(rc)
        with torch.cuda.device(g.device):
            _call("dsa_levdur_bwd", _p(g), _p(rc), _p(out), F, M, float(ctx.eps), _dtype_code(rc), _p(gr),
                  _stream())
        return gr, None


class LpcFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, M, eps):
        _require_device(x)
        xc = x.contiguous()
        L = xc.size(-1)
        F = xc.numel() // L
        out = torch.empty(*xc.shape[:-1], M + 1, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _call("dsa_lpc_fwd", _p(xc), F, L, M, float(eps), _dtype_code(xc), _p(out), _stream())
        ctx.save_for_backward(xc, out)
        ctx.cfg = (M, eps)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xc, out = ctx.saved_tensors
        M, eps = ctx.cfg
        g = g.contiguous()
        L = xc.size(-1)
        F = xc.numel() // L
        gx = torch.empty_like(xc)
        with torch.cuda.device(g.device):
            _call("dsa_lpc_bwd", _p(g), _p(xc), _p(out), F, L, M, float(eps), _dtype_code(xc), _p(gx), _stream())
        return gx, None, None


def frame_window_lpc(x, window, L, P, M, eps, center=True, mode="constant"):
    """Fused LPC branch (forward only): LPC(Window(Frame(x))), README.md:198-201 of the reference."""
    _require_device(x, window)
    _same_dtype(x, window)
    xc, wc = x.contiguous(), window.contiguous()
    T = xc.size(-1)
    B = xc.numel() // T
    out = torch.empty(*xc.shape[:-1], num_frames(T, P), M + 1, device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        _call("dsa_frame_window_lpc_fwd", _p(xc), B, T, L, P, _p(wc), int(center), pad_mode_code(mode), M,
              float(eps), _dtype_code(xc), _p(out), _stream())
    return out
