"""torch.autograd.Function wrappers over the C-ABI (include/diffsptk_amd.h).

PyTorch is plumbing here: it owns device memory and the stream; every computation below is a
call into libdiffsptk_amd.so.  Inputs must live on a HIP device -- there is deliberately no CPU
path (the reference's autograd-derived backward, SURVEY.md section 3.5, is replaced by the
hand-written backward kernels).
"""
from __future__ import annotations

import os
import weakref

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


def _scratch(device) -> torch.Tensor:
    """DSA_SCRATCH_BYTES of per-call workspace for the persistent tuned kernels (include/diffsptk_amd.h,
    Conventions): a fresh block from PyTorch's stream-ordered caching allocator, so calls that can overlap in
    time (other streams) never share one -- the library itself owns no device memory."""
    return torch.empty(_lib.SCRATCH_BYTES, dtype=torch.uint8, device=device)


_CLEAN_SCRATCH: dict = {}   # (device index, stream handle) -> a zeroed scratch the mel-cepstral forward keeps clean


def _clean_scratch(device) -> torch.Tensor:
    """A scratch that is zero on entry and left zero by the kernel (DSA_ALGO_SCRATCH_IS_CLEAN): one per (device, stream), so
    calls on one stream -- which cannot overlap -- share it and calls on different streams never do."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    t = _CLEAN_SCRATCH.get(key)
    if t is None:
        t = torch.zeros(_lib.SCRATCH_BYTES, dtype=torch.uint8, device=dev)
        _CLEAN_SCRATCH[key] = t
    return t


_IMAGES: dict = {}   # id(G) -> (weakref to G, versions, images): prepared operand images, made once per set of matrices


def mcep_images(G: torch.Tensor, D: torch.Tensor, E: torch.Tensor, fft_length: int, M: int):
    """The per-configuration constants of the tuned mel-cepstral kernels (dsa_mcep_prepare): binary16 hi/lo
    operand images of G, D, E, prepared once per set of matrices and reused by every call (None when the
    configuration has no tuned kernel).  Keyed by the tensors themselves (weakly) and their version counters, so
    moving a module to another device or editing a matrix in place prepares new images."""
    if G.device.type != "cuda" or G.dtype != torch.float32:
        return None
    lib = _lib.load()
    nbytes = lib.dsa_mcep_images_bytes(fft_length, M, _lib.F32)
    if nbytes <= 0:
        return None
    ver = (G._version, D._version, E._version, G.data_ptr(), D.data_ptr(), E.data_ptr())
    hit = _IMAGES.get(id(G))
    if hit is not None and hit[0]() is G and hit[1] == ver:
        if not hit[3].query():   # prepared on another stream and possibly not done yet
            with torch.cuda.device(G.device):
                torch.cuda.current_stream().wait_event(hit[3])
        return hit[2]
    Gc, Dc, Ec = G.contiguous(), D.contiguous(), E.contiguous()
    img = torch.empty(nbytes, dtype=torch.uint8, device=G.device)
    with torch.cuda.device(G.device):
        _call("dsa_mcep_prepare", _p(Gc), _p(Dc), _p(Ec), fft_length, M, _lib.F32, _p(img), _stream())
        # once per configuration; later calls may come from any stream, so they wait for this event on THEIR stream (no host
        # synchronisation: the preparation can sit inside a stream capture)
        ready = torch.cuda.Event()
        ready.record()
    if hit is None or hit[0]() is not G:
        weakref.finalize(G, _IMAGES.pop, id(G), None)   # the images go when the matrices go
    _IMAGES[id(G)] = (weakref.ref(G), ver, img, ready)
    return img


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
ROWS_PRO_LOG, ROWS_EPI_EXPSUB, ROWS_TRANS = 1, 2, 4   # DSA_ROWS_* (include/diffsptk_amd.h)


def rows_gemm(c, A, flags=0, aux=None):
    """out = op_out(op_in(c) @ B), B = A or (ROWS_TRANS) A^T: the library's general float32 row product on the matrix instruction
    (dsa_rows_gemm, csrc/rows_gemm.hip) with the fused log prologue / exp(aux - 2 .) epilogue of the untuned mel-cepstral step."""
    cc, Ac = c.contiguous(), A.contiguous()
    K = cc.size(-1)
    N = Ac.size(0) if flags & ROWS_TRANS else Ac.size(1)
    F = cc.numel() // K
    out = torch.empty(*cc.shape[:-1], N, device=c.device, dtype=c.dtype)
    auxc = None if aux is None else aux.contiguous()
    with torch.cuda.device(c.device):
        _call("dsa_rows_gemm", _p(cc), F, K, _p(Ac), Ac.size(1), N, flags, _p(auxc), N, _dtype_code(cc), _p(out), N, _stream())
    return out


def _row_product_is_long(Lin, Lout, elt) -> bool:
    """True where the library's row-product entry (csrc/mcep.hip:dsa_freqt_fwd / _bwd) would fall to its one-workgroup-per-row
    kernel: the matrix does not fit the LDS-resident kernel's 48 KB and the shape is outside the 257-bin matrix-core kernel's
    range -- the 1025-bin products of the 48 kHz set-ups.  Those run on the general matrix-core row product (rows_gemm).  The
    choice is a function of the GEOMETRY only (never of the number of rows): a frame's result does not depend on how many frames
    share its batch (tests/test_gpu_parity.py::test_row_products_are_batch_invariant)."""
    if os.environ.get("DSA_FREQT_GEMM", "1") == "0":
        return False
    lds_fits = elt * (Lin * Lout + 64 * (Lin + 1)) <= 48 * 1024
    return not lds_fits and max(Lin, Lout) >= 512


class MatmulRowsFn(torch.autograd.Function):
    """out = c @ A for a fixed (non-learnable) matrix A (freqt.py:141-143, mcep.py:286-288)."""

    @staticmethod
    def forward(ctx, c, A):
        _require_device(c, A)
        _same_dtype(c, A)
        cc, Ac = c.contiguous(), A.contiguous()
        L1, L2 = Ac.shape
        F = cc.numel() // L1
        ctx.save_for_backward(Ac)
        mfma = cc.dtype == torch.float32 and 48 < L1 <= 320 and L2 <= 192   # (the library picks its 257-bin matrix-core kernel)
        if not mfma and cc.dtype == torch.float32 and _row_product_is_long(L1, L2, cc.element_size()):
            return rows_gemm(cc, Ac)
        out = torch.empty(*cc.shape[:-1], L2, device=c.device, dtype=c.dtype)
        with torch.cuda.device(c.device):
            _call("dsa_freqt_fwd", _p(cc), F, L1, _p(Ac), L2, _dtype_code(cc), _p(out), _stream())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (Ac,) = ctx.saved_tensors
        g = g.contiguous()
        L1, L2 = Ac.shape
        F = g.numel() // L2
        if g.dtype == torch.float32 and _row_product_is_long(L2, L1, g.element_size()):
            return rows_gemm(g, Ac, ROWS_TRANS), None
        gc = torch.empty(*g.shape[:-1], L1, device=g.device, dtype=g.dtype)
        with torch.cuda.device(g.device):
            _call("dsa_freqt_bwd", _p(g), F, L1, _p(Ac), L2, _dtype_code(g), _p(gc), _stream())
        return gc, None


class RowsLogFn(torch.autograd.Function):
    """y = log(x) (mcep.py:203) as the library's own element-wise launch, differentiable (dsa_rows_ew op 0)."""

    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        y = torch.empty_like(xc)
        with torch.cuda.device(x.device):
            _call("dsa_rows_ew", 0, 0, _p(xc), None, None, xc.numel(), _dtype_code(xc), _p(y), None, _stream())
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (xc,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(xc)
        with torch.cuda.device(gy.device):
            _call("dsa_rows_ew", 0, 1, _p(xc), None, _p(gy), xc.numel(), _dtype_code(xc), _p(gx), None, _stream())
        return gx


class RowsExpSubFn(torch.autograd.Function):
    """y = exp(a - 2 b) (mcep.py:210-212), differentiable (dsa_rows_ew op 1; the backward needs the output only)."""

    @staticmethod
    def forward(ctx, a, b):
        ac, bc = a.contiguous(), b.contiguous()
        y = torch.empty_like(ac)
        with torch.cuda.device(a.device):
            _call("dsa_rows_ew", 1, 0, _p(ac), _p(bc), None, ac.numel(), _dtype_code(ac), _p(y), None, _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = gy.contiguous()
        ga, gb = torch.empty_like(y), torch.empty_like(y)
        with torch.cuda.device(gy.device):
            _call("dsa_rows_ew", 1, 1, _p(y), None, _p(gy), y.numel(), _dtype_code(y), _p(ga), _p(gb), _stream())
        return ga, gb


# ----------------------------------------------------------------------------------- inverse path (8(f) row 2)
# irfft = adjoint of rfft applied to c_k / N Y_k, overlap-add = adjoint of framing: the inverse ops run on
# the BACKWARD entry points of the analysis ops (and their gradients on the forward ones).
def _real_dtype(t):
    return {torch.complex64: torch.float32, torch.complex128: torch.float64}.get(t.dtype, t.dtype)


def _irfft_scale(yr, fft_length):
    """yr: (..., K, 2) real view of the half spectrum -> c_k / N * yr (ifftr.py:138)."""
    K = fft_length // 2 + 1
    out = torch.empty_like(yr)
    with torch.cuda.device(yr.device):
        _call("dsa_irfft_scale", _p(yr), yr.numel() // (2 * K), fft_length, _dtype_code(yr), _p(out), _stream())
    return out


def _div_rows(x2, d, eps=1e-16):
    out = torch.empty_like(x2)
    with torch.cuda.device(x2.device):
        _call("dsa_div_rows", _p(x2), x2.size(0), x2.size(1), _p(d), float(eps), _dtype_code(x2), _p(out), _stream())
    return out


def _fold_plan(N, L, P, center, out_length):
    """Signal length T the caller gets (unframe.py:176-192) and the length / frame count (Tc, Nc) the adjoint
    kernels are run with: Nc = num_frames(Tc) >= N (missing frames are zero), Tc >= T."""
    left = L // 2 if center else 0
    full = (N - 1) * P + L - left
    if out_length is None:
        T = N * P if center else full
    else:
        T = out_length
    T = max(0, min(T, full))        # slicing past the folded signal just ends there
    Tc = T if (T > 0 and num_frames(T, P) >= N) else max(T, (N - 1) * P + 1)
    return T, Tc, num_frames(Tc, P)


def _pad_frames(t, N, Nc, dim):
    if Nc == N:
        return t
    shape = list(t.shape)
    shape[dim] = Nc - N
    return torch.cat((t, t.new_zeros(shape)), dim=dim)


def _window_sq_sum(w, N, Nc, L, P, center, Tc):
    """Overlap-added squared window of the N frames, (Tc,): the divisor of unframe.py:204."""
    fr = (w * w).reshape(1, 1, L).expand(1, N, L)
    fr = _pad_frames(fr, N, Nc, 1).contiguous()
    d = torch.empty(1, Tc, device=w.device, dtype=w.dtype)
    with torch.cuda.device(w.device):
        _call("dsa_frame_bwd", _p(fr), 1, Tc, L, P, int(center), 0, 0, _dtype_code(fr), _p(d), _stream())
    return d.reshape(Tc)


class IfftrFn(torch.autograd.Function):
    """x:(..., out_length) = irfft(y:(..., L/2+1))[..., :out_length]  (ifftr.py:131-142)."""

    @staticmethod
    def forward(ctx, y, fft_length, out_length, twiddle):
        _require_device(y, twiddle)
        yr = torch.view_as_real(y.resolve_conj()).contiguous()
        _same_dtype(yr, twiddle)   # complex128 spectra need float64 tables: the kernels do not promote
        K = fft_length // 2 + 1
        F = yr.numel() // (2 * K)
        G = _irfft_scale(yr, fft_length)
        x0 = torch.zeros(F, out_length, device=y.device, dtype=yr.dtype)   # the adjoint is linear: any valid point
        x = torch.empty(*y.shape[:-1], out_length, device=y.device, dtype=yr.dtype)
        with torch.cuda.device(y.device):
            _call("dsa_fftr_bwd", _p(G), _p(x0), F, out_length, fft_length, 0, _p(twiddle), _dtype_code(yr), _p(x), _stream())
        ctx.save_for_backward(twiddle)
        ctx.cfg = (fft_length, out_length)
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, gx):
        (twiddle,) = ctx.saved_tensors
        fft_length, out_length = ctx.cfg
        gx = gx.contiguous()
        K = fft_length // 2 + 1
        F = gx.numel() // out_length
        Y = torch.empty(*gx.shape[:-1], K, 2, device=gx.device, dtype=gx.dtype)
        with torch.cuda.device(gx.device):
            _call("dsa_fftr_fwd", _p(gx), F, out_length, fft_length, 0, _p(twiddle), _dtype_code(gx), _p(Y), _stream())
        return torch.view_as_complex(_irfft_scale(Y, fft_length)), None, None, None


class UnframeFn(torch.autograd.Function):
    """x:(..., T) = overlap-add(y * w) / overlap-add(w^2)  (unframe.py:164-211); y:(..., N, L)."""

    @staticmethod
    def forward(ctx, y, w, P, center, out_length):
        _require_device(y, w)
        _same_dtype(y, w)
        if y.dim() <= 1:
            raise ValueError("Input must be at least 2D tensor.")
        yc, wc = y.contiguous(), w.contiguous()
        N, L = yc.shape[-2:]
        B = yc.numel() // (N * L)
        T, Tc, Nc = _fold_plan(N, L, P, center, out_length)
        yw = torch.empty_like(yc)
        num = torch.empty(B, Tc, device=y.device, dtype=y.dtype)
        with torch.cuda.device(y.device):
            _call("dsa_window_fwd", _p(yc), B * N, L, _p(wc), L, _dtype_code(yc), _p(yw), _stream())
            ywp = _pad_frames(yw.reshape(B, N, L), N, Nc, 1).contiguous()
            _call("dsa_frame_bwd", _p(ywp), B, Tc, L, P, int(center), 0, 0, _dtype_code(yc), _p(num), _stream())
        d = _window_sq_sum(wc, N, Nc, L, P, center, Tc)
        x = _div_rows(num, d)
        ctx.save_for_backward(wc, d)
        ctx.cfg = (yc.shape, P, center, T, Tc, Nc)
        return x[:, :T].reshape(*yc.shape[:-2], T)

    @staticmethod
    @once_differentiable
    def backward(ctx, gx):
        wc, d = ctx.saved_tensors
        shape, P, center, T, Tc, Nc = ctx.cfg
        N, L = shape[-2:]
        B = gx.numel() // max(T, 1)
        g2 = gx.reshape(B, T)
        if Tc != T:
            g2 = torch.cat((g2, g2.new_zeros(B, Tc - T)), dim=1)
        g2 = _div_rows(g2.contiguous(), d)
        fr = torch.empty(B, Nc, L, device=gx.device, dtype=gx.dtype)
        gy = torch.empty(B * N, L, device=gx.device, dtype=gx.dtype)
        with torch.cuda.device(gx.device):
            _call("dsa_frame_fwd", _p(g2), B, Tc, L, P, int(center), 0, 0, _dtype_code(g2), _p(fr), _stream())
            frn = fr[:, :N].contiguous()
            _call("dsa_window_fwd", _p(frn), B * N, L, _p(wc), L, _dtype_code(frn), _p(gy), _stream())
        return gy.reshape(shape), None, None, None, None


class IstftFn(torch.autograd.Function):
    """x:(..., T) = unframe(irfft(y)[..., :L])  (istft.py:186-193), fused: the complex-cotangent STFT backward
    kernel IS windowed inverse FFT + overlap-add."""

    @staticmethod
    def forward(ctx, y, window, twiddle, L, P, fft_length, center, out_length, algo):
        _require_device(y, window, twiddle)
        yr = torch.view_as_real(y.resolve_conj()).contiguous()
        _same_dtype(yr, window, twiddle)   # complex128 spectra need float64 tables: the kernels do not promote
        wc = window.contiguous()
        N, K = yr.shape[-3:-1]
        B = yr.numel() // (N * K * 2)
        T, Tc, Nc = _fold_plan(N, L, P, center, out_length)
        G = _pad_frames(yr.reshape(B, N, K, 2), N, Nc, 1).contiguous()
        d = _window_sq_sum(wc, N, Nc, L, P, center, Tc)
        x = torch.empty(B, Tc, device=y.device, dtype=yr.dtype)
        with torch.cuda.device(y.device):   # inverse weights while loading, overlap-add, division by d + 1e-16: one entry
            _call("dsa_istft_fwd", _p(G), B, Tc, L, P, fft_length, _p(wc), _p(twiddle), int(center), _p(d), 1e-16,
                  _dtype_code(yr), algo, _p(x), _stream())
        ctx.save_for_backward(wc, twiddle, d)
        ctx.cfg = (y.shape, L, P, fft_length, center, T, Tc, Nc, algo)
        return x[:, :T].reshape(*y.shape[:-2], T)

    @staticmethod
    @once_differentiable
    def backward(ctx, gx):
        wc, twiddle, d = ctx.saved_tensors
        shape, L, P, fft_length, center, T, Tc, Nc, algo = ctx.cfg
        N, K = shape[-2:]
        B = gx.numel() // max(T, 1)
        g2 = gx.reshape(B, T)
        if Tc != T:
            g2 = torch.cat((g2, g2.new_zeros(B, Tc - T)), dim=1)
        g2 = _div_rows(g2.contiguous(), d)
        Y = torch.empty(B, Nc, K, 2, device=gx.device, dtype=gx.dtype)
        with torch.cuda.device(gx.device):
            _call("dsa_stft_fwd", _p(g2), B, Tc, L, P, fft_length, _p(wc), _p(twiddle), int(center), 0, 0, 0.0, 0, 0.0, 5,
                  _dtype_code(g2), algo, _p(Y), _stream())   # format 5: complex output times c_k / nfft
        gy = Y[:, :N].contiguous()
        return (torch.view_as_complex(gy).reshape(shape),) + (None,) * 8


def griffin_update(t, y, phase, t_prev, d_prev, first, alpha, beta, gamma, eps, out=None):
    """One Griffin-Lim phase update (griffin.py:263-284, element-wise part): returns the next complex
    spectrogram sqrt(y + 1e-16) c / (|c| + eps); t_prev / d_prev (real views, (..., K, 2)) are updated in place.
    t=None is the initial step (phase=None: zeros)."""
    _require_device(y)
    _same_dtype(y, phase, t_prev, d_prev)
    K = y.size(-1)
    N = y.size(-2) if y.dim() >= 2 else 1
    B = y.numel() // max(N * K, 1)
    z = out if out is not None else torch.empty(*y.shape, dtype=torch.complex64 if y.dtype == torch.float32 else torch.complex128,
                                                device=y.device)
    zr = torch.view_as_real(z)
    tr, Nt = None, N
    if t is not None:
        if not t.is_complex() or t.size(-1) != K or t.size(-2) < N:
            raise ValueError("griffin_update: t must be the complex STFT of the current estimate")
        tr = torch.view_as_real(t.resolve_conj()).contiguous()
        Nt = t.size(-2)
    with torch.cuda.device(y.device):
        _call("dsa_griffin_update", _p(tr) if tr is not None else None, B, Nt, N, K, _p(y), _p(phase) if phase is not None else None,
              _p(t_prev), _p(d_prev), int(bool(first)), float(alpha), float(beta), float(gamma), float(eps), _dtype_code(y),
              _p(zr), _stream())
    return z


class FftcepFn(torch.autograd.Function):
    """CepstralAnalysis._forward (fftcep.py:116-136): x:(..., L/2+1) power spectra -> (..., M+1)."""

    @staticmethod
    def forward(ctx, x, A, cep_order, accel, n_iter):
        _require_device(x, A)
        _same_dtype(x, A)
        xc, Ac = x.contiguous(), A.contiguous()
        H = xc.size(-1)
        L = 2 * (H - 1)
        F = xc.numel() // H
        out = torch.empty(*xc.shape[:-1], cep_order + 1, device=x.device, dtype=x.dtype)
        masks = None
        if n_iter > 0 and x.requires_grad:
            masks = torch.empty(F, n_iter, (H + 63) // 64, device=x.device, dtype=torch.int64)
        with torch.cuda.device(x.device):
            _call("dsa_fftcep_fwd", _p(xc), F, L, cep_order, _p(Ac), float(accel), n_iter, _dtype_code(xc), _p(out),
                  _p(masks) if masks is not None else None, _stream())
        ctx.save_for_backward(xc, Ac, masks)
        ctx.cfg = (L, cep_order, float(accel), n_iter)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xc, Ac, masks = ctx.saved_tensors
        L, M, accel, n_iter = ctx.cfg
        H = xc.size(-1)
        F = xc.numel() // H
        gc = g.contiguous()
        gx = torch.empty_like(xc)
        with torch.cuda.device(g.device):
            _call("dsa_fftcep_bwd", _p(gc), _p(xc), F, L, M, _p(Ac), accel, n_iter, _p(masks) if masks is not None else None,
                  _dtype_code(xc), _p(gx), _stream())
        return gx, None, None, None, None


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


_FB_PLANS: dict = {}   # id(H) -> (weakref to H, version, plan or None)


def fbank_scan_plan(H: torch.Tensor):
    """The per-lane plan of the fused STFT -> filter-bank kernel for the (257, C) weights `H` (dsa_fbank_scan_plan: built
    on the host, one device-to-host copy of H per matrix and version), as a device tensor -- or None when H does not
    have the two-adjacent-channels-per-bin structure the kernel sums over (the two-kernel path serves those)."""
    import numpy as np

    key = id(H)
    hit = _FB_PLANS.get(key)
    if hit is not None and hit[0]() is H and hit[1] == H._version:
        return hit[2]
    plan = None
    if H.dim() == 2 and H.size(0) == 257 and 1 <= H.size(1) <= 126:
        Hh = np.ascontiguousarray(H.detach().to("cpu", torch.float64).numpy())
        table = np.zeros(_lib.FBANK_PLAN_FLOATS, dtype=np.float32)
        rc = _lib.load().dsa_fbank_scan_plan(Hh.ctypes.data, 257, int(H.size(1)), table.ctypes.data)
        if rc == 0:
            plan = torch.from_numpy(table).to(H.device)
        elif rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "dsa_fbank_scan_plan")
    _FB_PLANS[key] = (weakref.ref(H), H._version, plan)
    weakref.finalize(H, _FB_PLANS.pop, key, None)
    return plan


def stft_fbank(x, window, twiddle, L, P, fft_length, center, eps, plan, n_channel, floor, gamma, use_power):
    """y:(..., N, C) = glog(max(s H, floor)) of the STFT power values (or their square roots) in ONE launch
    (dsa_stft_fbank_fwd: stft.py:148-152 + fbank.py:306-321); no autograd graph (StftFbankFn wraps it with one)."""
    _require_device(x, window, twiddle, plan)
    _same_dtype(x, window, twiddle)
    xc, wc = x.contiguous(), window.contiguous()
    T = xc.size(-1)
    B = xc.numel() // T if T > 0 else 0
    y = torch.empty((*xc.shape[:-1], num_frames(T, P), n_channel), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        _call("dsa_stft_fbank_fwd", _p(xc), B, T, L, P, fft_length, _p(wc), _p(twiddle), int(center), float(eps), _p(plan),
              int(n_channel), float(floor), float(gamma), int(bool(use_power)), _dtype_code(xc), _p(y), _stream())
    return y


_BINS_TABLES: dict = {}   # id(H) -> (weakref to H, version, table or None)


def fbank_bins_table(H: torch.Tensor):
    """Device table of dsa_fbank_bins_bwd for the filter-bank matrix H (dsa_fbank_bins_plan: built on the host, once per
    matrix and version) -- or None when a bin feeds more than two adjacent channels."""
    import numpy as np

    key = id(H)
    hit = _BINS_TABLES.get(key)
    if hit is not None and hit[0]() is H and hit[1] == H._version:
        return hit[2]
    t = None
    if H.dim() == 2:
        Hh = np.ascontiguousarray(H.detach().to("cpu", torch.float64).numpy())
        table = np.zeros(4 * H.size(0), dtype=np.float32)
        if _lib.load().dsa_fbank_bins_plan(Hh.ctypes.data, int(H.size(0)), int(H.size(1)), table.ctypes.data) == 0:
            t = torch.from_numpy(table).to(H.device)
    if hit is None or hit[0]() is not H:
        weakref.finalize(H, _BINS_TABLES.pop, key, None)   # the table goes when the matrix goes
    _BINS_TABLES[key] = (weakref.ref(H), H._version, t)
    return t


class StftFbankFn(torch.autograd.Function):
    """stft_fbank with a gradient.  Forward: the one-launch kernel (the spectrogram never exists).  Backward, two launches:
    dsa_fbank_bins_bwd spreads the channel cotangents times d glog / d s (from the SAVED OUTPUT; 0 where the floor clamped, as
    torch.clip does in fbank.py:312) over the bins -- two multiply-adds per bin, the matrix has two entries per row -- and the
    result enters dsa_stft_bwd as the cotangent of the power / magnitude spectrum (which it recomputes from the waveform)."""

    @staticmethod
    def forward(ctx, x, window, twiddle, H, plan, L, P, fft_length, center, eps, floor, gamma, use_power):
        y = stft_fbank(x, window, twiddle, L, P, fft_length, center, eps, plan, H.size(1), floor, gamma, use_power)
        ctx.save_for_backward(x.contiguous(), window.contiguous(), twiddle, fbank_bins_table(H), y)
        ctx.cfg = (L, P, fft_length, center, eps, floor, gamma, use_power, H.size(0), H.size(1))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, wc, twiddle, table, y = ctx.saved_tensors
        L, P, fft_length, center, eps, floor, gamma, use_power, K, Cn = ctx.cfg
        gy = gy.contiguous()
        F = gy.numel() // Cn
        g = torch.empty(*gy.shape[:-1], K, device=gy.device, dtype=gy.dtype)
        T = xc.size(-1)
        B = xc.numel() // T
        gx = torch.empty_like(xc)
        with torch.cuda.device(gy.device):
            _call("dsa_fbank_bins_bwd", _p(gy), _p(y), F, K, Cn, _p(table), float(floor), float(gamma), _dtype_code(gy), _p(g),
                  _stream())
            _call("dsa_stft_bwd", _p(g), _p(xc), B, T, L, P, fft_length, _p(wc), _p(twiddle), int(center), 0,
                  pad_mode_code("constant"), float(eps), 0, 0.0, 3 if use_power else 2, _dtype_code(xc), _lib.ALGO_AUTO,
                  _p(gx), None, _stream())
        return (gx,) + (None,) * 12


class MfccFn(torch.autograd.Function):
    """cy, E = (glog(max(s H, floor)) W, log energy) in one launch (mfcc.py:244-256): W:(C, M+1) is DCT-II x truncation
    x liftering vector.  Backward = the transposed product through W, then the filter-bank backward."""

    @staticmethod
    def forward(ctx, x, H, W, floor, gamma, use_power):
        _require_device(x, H, W)
        _same_dtype(x, H)
        _same_dtype(x, W)
        xc, Hc, Wc = x.contiguous(), H.contiguous(), W.contiguous()
        K, Cn = Hc.shape
        Mo = Wc.size(1)
        F = xc.numel() // K
        z = torch.empty(*xc.shape[:-1], Mo, device=x.device, dtype=x.dtype)
        E = torch.empty(*xc.shape[:-1], 1, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _call("dsa_fbank_dct_fwd", _p(xc), F, K, _p(Hc), Cn, _p(Wc), Mo, float(floor), float(gamma), int(bool(use_power)),
                  _dtype_code(xc), _p(z), _p(E), _stream())
        ctx.save_for_backward(xc, Hc, Wc)
        ctx.cfg = (float(floor), float(gamma), int(bool(use_power)))
        return z, E

    @staticmethod
    @once_differentiable
    def backward(ctx, gz, gE):
        xc, Hc, Wc = ctx.saved_tensors
        floor, gamma, use_power = ctx.cfg
        K, Cn = Hc.shape
        Mo = Wc.size(1)
        F = xc.numel() // K
        gy = torch.empty(*xc.shape[:-1], Cn, device=xc.device, dtype=xc.dtype)
        gx = torch.empty_like(xc)
        gEc = gE.contiguous() if gE is not None else None
        with torch.cuda.device(xc.device):
            if gz is not None:
                _call("dsa_freqt_bwd", _p(gz.contiguous()), F, Cn, _p(Wc), Mo, _dtype_code(xc), _p(gy), _stream())
            else:
                gy.zero_()
            _call("dsa_fbank_bwd", _p(gy), _p(gEc) if gEc is not None else None, _p(xc), F, K, _p(Hc), Cn, floor, gamma,
                  use_power, _dtype_code(xc), _p(gx), _stream())
        return gx, None, None, None, None, None


# ----------------------------------------------------------------------------------- mcep
def _mcep_composed_applies(Xc, M) -> bool:
    """Geometries without a tuned kernel (48 kHz set-ups: fft_length 1024 / 2048, orders 34 .. 60).  A function of the geometry
    and the dtype only -- never of the number of frames: a frame's mel-cepstrum does not depend on how many frames share its
    batch.  Short spectra (fft_length < 256: the reference's own test grids) keep the generic kernel pair."""
    return (Xc.dtype == torch.float32 and M + 1 <= 64 and M >= 1 and Xc.size(-1) >= 129
            and os.environ.get("DSA_MCEP_COMPOSED", "1") != "0")   # float64 keeps the generic kernel pair


def mcep_composed(X, G, D, E, av, fft_length, M, n_iter, algo):
    """The mel-cepstral analysis for a geometry without a tuned kernel, WITH a graph when one is wanted: the whole-batch
    launches of _mcep_composed_fwd are differentiable operations (GEMMs, element-wise, ThSolveFn), so autograd runs the
    backward as whole-batch launches too (the generic kernel pair keeps one workgroup per frame in both directions).
    None: not applicable (a tuned kernel exists, the generic family was asked for, or the spectrum is short)."""
    if algo == _lib.ALGO_GENERIC or X.device.type != "cuda":
        return None
    K = fft_length // 2 + 1
    F = X.numel() // K
    if not _mcep_composed_applies(X, M) or mcep_images(G, D, E, fft_length, M) is not None:
        return None
    _require_device(X, G, D, E, av)
    _same_dtype(X, G, D, E, av)
    return _mcep_composed_fwd(X.contiguous(), G, D, E, av, M, n_iter)


def mcep_newton_update(rt, av, mc):
    """mc + solve(T(rt[:, :n]) + H(rt), rt[:, :n] - av) (mcep.py:216-222; dsa_mcep_newton_update, float32, n <= 55)."""
    n = mc.size(-1)
    out = torch.empty_like(mc)
    with torch.cuda.device(mc.device):
        _call("dsa_mcep_newton_update", _p(rt), mc.numel() // n, n, _p(av), _dtype_code(mc), _p(mc), _p(out), _stream())
    return out


def gnorm(x, gamma, inverse=False):
    """Gain normalisation (gnorm.py:102-112) / its inverse (ignorm.py:99-109) of (..., M + 1) rows in one launch (dsa_gnorm_fwd);
    forward only -- the modules keep the stock composition when a gradient is wanted."""
    xc = x.contiguous()
    n = xc.size(-1)
    out = torch.empty_like(xc)
    with torch.cuda.device(x.device):
        _call("dsa_gnorm_fwd", _p(xc), xc.numel() // n, n, float(gamma), int(bool(inverse)), _dtype_code(xc), _p(out), _stream())
    return out


def gnorm_applies(x) -> bool:
    """dsa_gnorm_fwd takes this call: a device tensor in float32 / float64 and no gradient wanted."""
    return x.is_cuda and x.dtype in (torch.float32, torch.float64) and not (torch.is_grad_enabled() and x.requires_grad) and x.numel() > 0


def mgcep_gain(r, b_eps, gamma, b_join):
    """(sqrt(r_0 + gamma sum_m r_{m+1} b_eps_m), b_join) as one (..., M + 1) tensor (mgcep.py:213-215, 221, 231-233; dsa_mgcep_gain),
    forward only."""
    rc, bc, jc = r.contiguous(), b_eps.contiguous(), b_join.contiguous()
    M = bc.size(-1)
    out = torch.empty(*bc.shape[:-1], M + 1, device=bc.device, dtype=bc.dtype)
    with torch.cuda.device(bc.device):
        _call("dsa_mgcep_gain", _p(rc), _p(bc), _p(jc), bc.numel() // M, M, float(gamma), _dtype_code(bc), _p(out), _stream())
    return out


class McepNewtonUpdateFn(torch.autograd.Function):
    """mc + solve(T(rt[:, :n]) + H(rt), rt[:, :n] - av) with a gradient (mcep.py:216-222): forward = the batched solve (the solution is
    kept), backward = the same solve on the cotangent and one launch of diagonal sums (dsa_mcep_newton_update_bwd).  As a slice, a
    subtraction, ThSolveFn and an addition the step was nine small stock launches around the two kernels, in each direction."""

    @staticmethod
    def forward(ctx, rt, av, mc):
        rtc = rt.contiguous()
        n = mc.size(-1)
        sol = torch.empty_like(mc)
        with torch.cuda.device(mc.device):
            _call("dsa_mcep_newton_update", _p(rtc), mc.numel() // n, n, _p(av), _dtype_code(mc), None, _p(sol), _stream())
        ctx.save_for_backward(rtc, sol)
        return mc + sol

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        rtc, sol = ctx.saved_tensors
        n = sol.size(-1)
        gc = g.contiguous()
        u, grt = torch.empty_like(sol), torch.empty_like(rtc)
        with torch.cuda.device(g.device):
            _call("dsa_mcep_newton_update_bwd", _p(gc), _p(rtc), _p(sol), sol.numel() // n, n, _dtype_code(sol), _p(u), _p(grt), _stream())
        return grt, None, g


def mcep_newton_resid(logx, mc, D, E):
    """rt = exp(logx - 2 mc D) E (mcep.py:210-215) in one launch (dsa_mcep_newton_resid: float32, 3 <= M + 1 <= 55): e is formed
    chunk by chunk in the operand layout of the second product and never reaches memory."""
    n, K = mc.size(-1), logx.size(-1)
    Dc, Ec = D.contiguous(), E.contiguous()
    rt = torch.empty(*mc.shape[:-1], 2 * n - 1, device=mc.device, dtype=mc.dtype)
    with torch.cuda.device(mc.device):
        _call("dsa_mcep_newton_resid", _p(logx), mc.numel() // n, K, _p(mc), n, _p(Dc), Dc.size(1), _p(Ec), Ec.size(1),
              _dtype_code(mc), _p(rt), _stream())
    return rt


_RESID_IMAGES: dict = {}   # id(D) -> (weakref to D, versions, images, ready event): the binary16 operand images of dsa_mcep_newton_resid_h


def mcep_resid_images(D, E):
    """The binary16 hi / lo operand images dsa_mcep_newton_resid_h consumes (dsa_mcep_resid_prepare: one small launch), made once per
    pair of tables and kept as long as the tables live (keyed like mcep_images: weakly by the tensors and their version counters; a call
    from another stream waits for the preparation's event on ITS stream)."""
    if D.device.type != "cuda" or D.dtype != torch.float32 or E.dtype != torch.float32:
        return None
    n, K = D.size(0), D.size(1)
    nbytes = _lib.load().dsa_mcep_resid_images_bytes(K, n)
    if nbytes <= 0:
        return None
    ver = (D._version, E._version, D.data_ptr(), E.data_ptr(), tuple(D.shape), tuple(E.shape))
    hit = _RESID_IMAGES.get(id(D))
    if hit is not None and hit[0]() is D and hit[1] == ver:
        if not hit[3].query():
            with torch.cuda.device(D.device):
                torch.cuda.current_stream().wait_event(hit[3])
        return hit[2]
    Dc, Ec = D.contiguous(), E.contiguous()
    images = torch.empty(nbytes, dtype=torch.uint8, device=D.device)
    with torch.cuda.device(D.device):
        _call("dsa_mcep_resid_prepare", _p(Dc), Dc.size(1), _p(Ec), Ec.size(1), K, n, _dtype_code(Dc), _p(images), _stream())
        ready = torch.cuda.Event()
        ready.record()
    if hit is None or hit[0]() is not D:
        weakref.finalize(D, _RESID_IMAGES.pop, id(D), None)
    _RESID_IMAGES[id(D)] = (weakref.ref(D), ver, images, ready)
    return images


_RESID_BWD_IMAGES: dict = {}   # id(D) -> (weakref to D, versions, images, ready event): the operand images of dsa_mcep_newton_resid_h_bwd


def mcep_resid_bwd_images(D, E):
    """The binary16 hi / lo operand images dsa_mcep_newton_resid_h_bwd consumes (dsa_mcep_resid_bwd_prepare), made once per pair of
    tables and kept as long as the tables live (as mcep_resid_images); None where no kernel covers the order."""
    if D.device.type != "cuda" or D.dtype != torch.float32 or E.dtype != torch.float32:
        return None
    n, K = D.size(0), D.size(1)
    nbytes = _lib.load().dsa_mcep_resid_bwd_images_bytes(K, n)
    if nbytes <= 0:
        return None
    ver = (D._version, E._version, D.data_ptr(), E.data_ptr(), tuple(D.shape), tuple(E.shape))
    hit = _RESID_BWD_IMAGES.get(id(D))
    if hit is not None and hit[0]() is D and hit[1] == ver:
        if not hit[3].query():
            with torch.cuda.device(D.device):
                torch.cuda.current_stream().wait_event(hit[3])
        return hit[2]
    Dc, Ec = D.contiguous(), E.contiguous()
    images = torch.empty(nbytes, dtype=torch.uint8, device=D.device)
    with torch.cuda.device(D.device):
        _call("dsa_mcep_resid_bwd_prepare", _p(Dc), Dc.size(1), _p(Ec), Ec.size(1), K, n, _dtype_code(Dc), _p(images), _stream())
        ready = torch.cuda.Event()
        ready.record()
    if hit is None or hit[0]() is not D:
        weakref.finalize(D, _RESID_BWD_IMAGES.pop, id(D), None)
    _RESID_BWD_IMAGES[id(D)] = (weakref.ref(D), ver, images, ready)
    return images


MCEP_GLOGX_ONE_PASS = True   # McepNewtonStepsHFn.backward: glogx in one pass after the sweep (dsa_mcep_newton_glogx_h)
MCEP_GLOGX_MIN_FRAMES = 20480   # (tools/sweep_glogx_threshold.py: 16 384 frames 2.55 against 2.53 ms accumulating, 24 576: 3.40 against 3.73)


class McepNewtonStepsHFn(torch.autograd.Function):
    """mcep.py:208-222 at the 48 kHz set-ups (orders 32 .. 54) WITH a gradient, as one node (round 6): forward = per Newton step
    dsa_mcep_newton_resid_h + dsa_mcep_newton_update, the iterates, rt rows and solutions kept ((3 n + ...) floats per frame and step
    -- not e:(F, K)); backward = per step, in reverse, dsa_mcep_newton_update_bwd (the solve on the cotangent, the diagonal sums) and
    dsa_mcep_newton_resid_h_bwd (e recomputed from the iterate; glogx accumulated in place).  Inputs: logx (natural logarithms of the
    spectrum) and the start mc0 = logx G; the tables D, E, alpha_vec carry no gradient here (a learnable basis takes the composed
    path).  Replaces, per step and 102 400 frames at 2048 / 49, 2.0 ms of differentiable pieces by 0.5 + 0.7 ms."""

    @staticmethod
    def forward(ctx, logx, mc0, D, E, av, n_iter):
        images = mcep_resid_images(D, E)
        F, n = mc0.numel() // mc0.size(-1), mc0.size(-1)
        K = logx.size(-1)
        lx = logx.contiguous()
        mcs = torch.empty(n_iter + 1, F, n, device=mc0.device, dtype=mc0.dtype)   # the iterates
        rts = torch.empty(n_iter, F, 2 * n - 1, device=mc0.device, dtype=mc0.dtype)
        sols = torch.empty(n_iter, F, n, device=mc0.device, dtype=mc0.dtype)
        mcs[0].copy_(mc0.reshape(F, n))
        with torch.cuda.device(mc0.device):
            for i in range(n_iter):
                _call("dsa_mcep_newton_resid_h", _p(lx), F, K, _p(mcs[i]), n, _p(images), _dtype_code(lx), _p(rts[i]), _stream())
                _call("dsa_mcep_newton_update", _p(rts[i]), F, n, _p(av), _dtype_code(lx), None, _p(sols[i]), _stream())
                torch.add(mcs[i], sols[i], out=mcs[i + 1])
        ctx.save_for_backward(lx, mcs, rts, sols, D, E)
        ctx.n_iter = n_iter
        return mcs[n_iter].reshape(mc0.shape).clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lx, mcs, rts, sols, D, E = ctx.saved_tensors
        n_iter = ctx.n_iter
        F, n = mcs.size(1), mcs.size(2)
        K = lx.size(-1)
        images_b = mcep_resid_bwd_images(D, E)
        gbar = g.reshape(F, n).contiguous().clone()
        u, gmc = torch.empty_like(gbar), torch.empty_like(gbar)
        # 0.2.2: the sum over the steps that lands in glogx is formed AFTER the sweep, in one pass over the bins from the iterates and the
        # steps' cotangents grt (kept: 4 (2 n - 1) bytes per frame and step) -- dsa_mcep_newton_glogx_h; the sweep's launches then leave
        # glogx alone (a step moves the (F, K) array once instead of three times).  Same values summed in the same order: the same bits
        # as the in-place accumulation (DSA_MCEP_GLOGX_PASS=0, and n_iter beyond what the pass holds on chip).
        # (from MCEP_GLOGX_MIN_FRAMES frames on: below, the pass's one-tile workgroups leave CUs idle -- 12 800 frames at 2048 / 49: 2.42 ms
        #  accumulating, 2.48 with the pass; DSA_MCEP_GLOGX_PASS=1 forces it, =0 forbids it.  The bits do not depend on the choice.)
        env = os.environ.get("DSA_MCEP_GLOGX_PASS", "")
        one_pass = (MCEP_GLOGX_ONE_PASS and env != "0" and (F >= MCEP_GLOGX_MIN_FRAMES or env == "1")
                    and n_iter * ((4 + 2 * ((2 * n - 1 + 31) // 32)) * 1024 + 128) <= 156 * 1024)
        with torch.cuda.device(g.device):
            if one_pass:
                grts = torch.empty_like(rts)
                for i in range(n_iter - 1, -1, -1):
                    _call("dsa_mcep_newton_update_bwd", _p(gbar), _p(rts[i]), _p(sols[i]), F, n, _dtype_code(lx), _p(u), _p(grts[i]), _stream())
                    _call("dsa_mcep_newton_resid_h_bwd", _p(lx), F, K, _p(mcs[i]), n, _p(grts[i]), _p(images_b), _dtype_code(lx), None, _p(gmc),
                          _stream())
                    gbar.add_(gmc)
                glogx = torch.empty_like(lx)
                rc = getattr(_lib.load(), "dsa_mcep_newton_glogx_h")(_p(lx), F, K, _p(mcs), n, _p(grts), int(n_iter), _p(images_b),
                                                                       _dtype_code(lx), _p(glogx), _stream())
                if rc == _lib.ERR_UNSUPPORTED:   # (cannot happen for what mcep_newton_steps_grad_applies admits; kept as the contract says)
                    glogx.zero_()
                    for i in range(n_iter):
                        _call("dsa_mcep_newton_resid_h_bwd", _p(lx), F, K, _p(mcs[i]), n, _p(grts[i]), _p(images_b), _dtype_code(lx), _p(glogx),
                              _p(gmc), _stream())
                else:
                    _lib.check(rc, "dsa_mcep_newton_glogx_h")
            else:
                glogx = torch.zeros_like(lx)
                grt = torch.empty_like(rts[0])
                for i in range(n_iter - 1, -1, -1):
                    _call("dsa_mcep_newton_update_bwd", _p(gbar), _p(rts[i]), _p(sols[i]), F, n, _dtype_code(lx), _p(u), _p(grt), _stream())
                    _call("dsa_mcep_newton_resid_h_bwd", _p(lx), F, K, _p(mcs[i]), n, _p(grt), _p(images_b), _dtype_code(lx), _p(glogx), _p(gmc),
                          _stream())
                    gbar.add_(gmc)
        return glogx.reshape(lx.shape), gbar.reshape(g.shape), None, None, None, None


def mcep_newton_steps_grad_applies(M1, D, E, av):
    """McepNewtonStepsHFn takes the analysis: orders 32 .. 54, float32 tables without a gradient of their own
    (DSA_MCEP_GRAD_H=0: the composed path, for A/B runs)."""
    return (33 <= M1 <= 55 and os.environ.get("DSA_MCEP_GRAD_H", "1") != "0" and D.dtype == torch.float32 and E.dtype == torch.float32
            and av.dtype == torch.float32 and not (D.requires_grad or E.requires_grad or av.requires_grad)
            and D.device.type == "cuda" and _lib.load().dsa_mcep_resid_bwd_images_bytes(D.size(1), D.size(0)) > 0
            and _lib.load().dsa_mcep_resid_images_bytes(D.size(1), D.size(0)) > 0)


def mcep_newton_resid_h(logx, mc, images):
    """rt = exp(logx - 2 mc D) E (mcep.py:210-215) in one launch with both products as 3-term binary16 splits on the matrix pipe
    (dsa_mcep_newton_resid_h: float32, 3 <= M + 1 <= 55); `images` from mcep_resid_images(D, E)."""
    n, K = mc.size(-1), logx.size(-1)
    rt = torch.empty(*mc.shape[:-1], 2 * n - 1, device=mc.device, dtype=mc.dtype)
    with torch.cuda.device(mc.device):
        _call("dsa_mcep_newton_resid_h", _p(logx), mc.numel() // n, K, _p(mc), n, _p(images), _dtype_code(mc), _p(rt), _stream())
    return rt


def mcep_newton_steps_applies(M1: int) -> bool:
    """dsa_mcep_newton_steps has an instantiation for this order (32 .. 54, among them the 48 kHz set-ups fft_length 2048 / order 49
    and 1024 / order 34; DSA_MCEP_BIG=0: the two-launch step, for A/B runs)."""
    return 33 <= M1 <= 55 and os.environ.get("DSA_MCEP_BIG", "1") != "0"


def mcep_newton_steps(logx, mc0, images, av, n_iter):
    """ALL n_iter Newton steps of mcep.py:208-222 in ONE persistent launch (dsa_mcep_newton_steps, csrc/mcep_big_f16.h): per step the
    products of mcep_newton_resid_h and the solve-and-update of mcep_newton_update, rt and mc staying on chip; forward only.
    None: the library has no instantiation for this order after all (the caller runs the two launches per step)."""
    n, K = mc0.size(-1), logx.size(-1)
    mcc = mc0.contiguous()
    out = torch.empty_like(mcc)
    with torch.cuda.device(mcc.device):
        rc = getattr(_lib.load(), "dsa_mcep_newton_steps")(_p(logx), mcc.numel() // n, K, _p(mcc), n, _p(images), _p(av), int(n_iter),
                                                            _dtype_code(mcc), _p(out), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "dsa_mcep_newton_steps")
    return out


def _mcep_composed_fwd(Xc, G, D, E, av, M, n_iter):
    """mcep.py:203-222 for the geometries the tuned kernel does not cover, as whole-batch launches of the library's own kernels
    instead of the one-workgroup-per-frame generic kernel: per Newton step the two row products (F, M+1) x (M+1, K) and
    (F, K) x (K, 2M+1) with exp(log X - 2 .) between them as ONE matrix-core launch (dsa_mcep_newton_resid; orders above 54: two
    launches of dsa_rows_gemm) and the batched Toeplitz-plus-Hankel solve-and-update (dsa_mcep_newton_update).  With a
    graph wanted the same composition runs on differentiable pieces (RowsLogFn, MatmulRowsFn, RowsExpSubFn, ThSolveFn).  Same
    arithmetic order per frame as the reference's formulation; float32 products accumulate in float32."""
    M1 = M + 1
    lead = Xc.shape[:-1]
    X2 = Xc.reshape(-1, Xc.size(-1))
    want_grad = torch.is_grad_enabled() and X2.requires_grad
    if X2.dtype != torch.float32:
        raise _lib.BackendError("mcep (whole-batch composition): float32 only")
    if want_grad:
        logx = RowsLogFn.apply(X2)                                        # mcep.py:203
        mc = MatmulRowsFn.apply(logx, G)                                  # :204-207
    else:
        logx = RowsLogFn.apply(X2)                                        # kept: every step's epilogue reads it
        mc = rows_gemm(logx, G)
    one_launch_resid = os.environ.get("DSA_MCEP_RESID", "1") != "0" and Xc.size(-1) >= 4
    # round 5: the step's two products as binary16 splits (DSA_MCEP_RESID_H=0: the float32 matrix instructions of round 4, for A/B runs)
    images_h = None
    if not want_grad and 3 <= M1 <= 55 and one_launch_resid and os.environ.get("DSA_MCEP_RESID_H", "1") != "0" \
            and D.dtype == torch.float32 and E.dtype == torch.float32:
        images_h = mcep_resid_images(D, E)
    if images_h is not None and n_iter >= 1 and mcep_newton_steps_applies(M1) and av.dtype == torch.float32:
        # round 6: every step in one persistent launch (22 launches -> 3 for the analysis)
        out_ = mcep_newton_steps(logx, mc, images_h, av, n_iter)
        if out_ is not None:
            return out_.reshape(*lead, M1)
    if want_grad and n_iter >= 1 and Xc.size(-1) >= 4 and mcep_newton_steps_grad_applies(M1, D, E, av):
        # round 6: the steps as ONE node whose backward is two launches per step (dsa_mcep_newton_update_bwd, dsa_mcep_newton_resid_h_bwd)
        return McepNewtonStepsHFn.apply(logx, mc, D, E, av, n_iter).reshape(*lead, M1)
    for _ in range(n_iter):
        if want_grad:
            e = RowsExpSubFn.apply(logx, MatmulRowsFn.apply(mc, D))       # :210-212
            rt = MatmulRowsFn.apply(e, E)                                 # :214-215
        elif images_h is not None:
            rt = mcep_newton_resid_h(logx, mc.contiguous(), images_h)     # :210-215 in one launch on the binary16 matrix pipe
        elif 3 <= M1 <= 55 and one_launch_resid:
            rt = mcep_newton_resid(logx, mc, D, E)                        # :210-215 in one launch, e never stored
        else:
            e = rows_gemm(mc, D, ROWS_EPI_EXPSUB, aux=logx)               # product and exp(log X - 2 .) in one launch
            rt = rows_gemm(e, E)
        if want_grad and 2 <= M1 <= 55:
            mc = McepNewtonUpdateFn.apply(rt, av, mc)                     # :216-222, one node
        elif want_grad or M1 > 55:
            p = rt[:, :M1].contiguous()
            mc = mc + ThSolveFn.apply(p, rt, p - av)                      # :216-222
        else:
            mc = mcep_newton_update(rt, av, mc)                           # the same in one launch: 16 systems per wave
    return mc.reshape(*lead, M1)


# The tuned mel-cepstral forward can keep every Newton step's (2 M + 1)-entry row of rt behind the iterates a gradient needs anyway:
# the backward then skips its second forward chain (1.56 -> 1.09-1.18 ms per 204 800 frames).  Cost: n_iter F (2 M + 1) floats on top
# of the (n_iter + 1) F (M + 1) of the iterates -- at 204 800 frames and 10 steps 401 MB on top of 225 MB, alive from the forward to
# the backward.  Above this many EXTRA bytes per call the rows are not kept and the backward recomputes them (same gradient to
# rounding; slower); set it to 0 to never keep them, or DSA_MCEP_HIST_RT=0 in the environment.
MCEP_HIST_RT_MAX_BYTES = 4 << 30


def _keep_rt_rows(n_iter, F, M, like):
    if os.environ.get("DSA_MCEP_HIST_RT", "1") == "0":
        return False
    return n_iter * F * (2 * M + 1) * like.element_size() <= MCEP_HIST_RT_MAX_BYTES


def _mcep_history(n_iter, F, M, like, with_rt):
    """The Newton history a gradient needs: (n_iter + 1, F, M + 1) iterates, followed -- for the tuned kernels (with_rt) -- by the
    (n_iter, F, 2 M + 1) rows of rt that let the backward skip its second forward chain (DSA_ALGO_HIST_HAS_RT).  One flat buffer."""
    n = (n_iter + 1) * F * (M + 1) + (n_iter * F * (2 * M + 1) if with_rt else 0)
    return torch.empty(n, device=like.device, dtype=like.dtype)


_overlapped = [False]


class overlapped_launches:
    """``with ops.overlapped_launches():`` -- the caller alternates consecutive, independent analysis calls between two streams
    (bench.py --streams 2, dist.analyze_chunked_overlap(alternate_streams=True)).  The tuned mel-cepstral forward launches then pack
    their short last round of tiles onto a few workgroups and release every other CU to the next launch, which waits on the other
    stream (DSA_ALGO_OVERLAPPED_LAUNCHES, include/diffsptk_amd.h): 6.25 rounds per 204 800 frames in the steady state instead of
    6.8.  Results are bit-identical either way; a lone launch is slower with it, so it is never the default."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _overlapped[0]
        _overlapped[0] = self.on
        return self

    def __exit__(self, *exc):
        _overlapped[0] = self.prev
        return False


_reserved_cus = [0]


class reserve_cus:
    """``with ops.reserve_cus(n):`` -- the tuned mel-cepstral forward launches (STFT -> mel-cepstrum in one launch, mcep alone) leave
    ``n`` of the 256 CUs free (DSA_ALGO_RESERVE_CUS, include/diffsptk_amd.h).  A persistent workgroup fills its CU, so a kernel of
    another stream -- RCCL's all-gather of the previous batch's features -- otherwise starts only in the launch's tail and the next
    launch queues behind it.  dist.analyze_chunked_overlap sets it in a world of more than one rank.  Same bits; the launch itself
    takes 256 / (256 - n) as long."""

    def __init__(self, n: int):
        self.n = max(0, min(63, int(n)))

    def __enter__(self):
        self.prev = _reserved_cus[0]
        _reserved_cus[0] = self.n
        return self

    def __exit__(self, *exc):
        _reserved_cus[0] = self.prev
        return False


def _mcep_scratch(device):
    """(scratch, algo flag) of a tuned mel-cepstral forward launch: the per-(device, stream) kept-zero counters
    (DSA_ALGO_SCRATCH_IS_CLEAN, no fill launch per call) -- except while a HIP graph is being captured: a graph replays on whatever
    stream is current, possibly next to an eager call that uses the capture stream's counters, so a captured launch gets its
    own block and the library's reset (a captured memset node) instead."""
    with torch.cuda.device(device):
        if torch.cuda.is_current_stream_capturing() or os.environ.get("DSA_CLEAN_SCRATCH", "1") == "0":   # (the variable: A/B runs)
            return _scratch(device), _lib.algo_reserve_cus(_reserved_cus[0])
        return _clean_scratch(device), (_lib.ALGO_SCRATCH_IS_CLEAN | (_lib.ALGO_OVERLAPPED_LAUNCHES if _overlapped[0] else 0)
                                        | _lib.algo_reserve_cus(_reserved_cus[0]))


def stft_mcep_fusable(x, window, G, L, P, fft_length, M) -> bool:
    """Configurations dsa_stft_mcep_fwd covers (include/diffsptk_amd.h): float32 device tensors, frame_length 400, fft_length 512,
    cep_order 24 (the caller checks power format / constant padding / no zmean / no relative floor)."""
    if not (x.is_cuda and x.dtype == torch.float32 and window.dtype == torch.float32 and G.dtype == torch.float32):
        return False
    T = x.size(-1)
    if T < 1 or T >= 2 ** 31 or L != 400 or fft_length != 512 or M != 24 or P < 1:
        return False
    B = x.numel() // T
    return B * num_frames(T, P) < 2 ** 31


class StftMcepFn(torch.autograd.Function):
    """MelCepstralAnalysis(STFT(x)) in ONE launch (dsa_stft_mcep_fwd; stft.py:237-241 -> mcep.py:189-224): the (B, N, 257) power
    spectrogram is neither written nor re-read -- unless a gradient is wanted: then the same launch also leaves the spectrogram
    and the Newton history behind, and the backward is the two stages' own (dsa_mcep_bwd, then dsa_stft_bwd)."""

    @staticmethod
    def forward(ctx, x, window, twiddle, G, D, E, av, L, P, fft_length, center, eps, M, n_iter, mode="constant", zmean=False,
                relative_floor_db=None):
        _require_device(x, window, twiddle, G, D, E, av)
        _same_dtype(x, window, twiddle, G, D, E, av)
        xc, wc = x.contiguous(), window.contiguous()
        T = xc.size(-1)
        B = xc.numel() // T
        N = num_frames(T, P)
        K = fft_length // 2 + 1
        F = B * N
        need_grad = ctx.needs_input_grad[0]
        mc = torch.empty(*xc.shape[:-1], N, M + 1, device=x.device, dtype=x.dtype)
        with_rt = need_grad and _keep_rt_rows(n_iter, F, M, xc)
        hist = _mcep_history(n_iter, F, M, x, with_rt) if need_grad else None
        X = torch.empty(*xc.shape[:-1], N, K, device=x.device, dtype=x.dtype) if need_grad else None
        images = mcep_images(G, D, E, fft_length, M)
        if images is None:
            raise _lib.BackendError("stft_mcep: no tuned kernel for this configuration (check stft_mcep_fusable first)")
        scratch, flag = _mcep_scratch(x.device)
        if with_rt:
            flag |= _lib.ALGO_HIST_HAS_RT
        with torch.cuda.device(x.device):
            _call("dsa_stft_mcep_opts_fwd", _p(xc), B, T, L, P, fft_length, _p(wc), _p(twiddle), int(center), int(bool(zmean)),
                  pad_mode_code(mode), float(eps), int(relative_floor_db is not None),
                  0.0 if relative_floor_db is None else float(relative_floor_db), M, n_iter, _p(G), _p(D), _p(E), _p(av), _dtype_code(xc),
                  _lib.ALGO_AUTO | flag, _p(images), _p(scratch), _p(mc), _p(hist), _p(X), _stream())
        if need_grad:
            ctx.save_for_backward(xc, wc, twiddle, X, hist, G, D, E, av)
        ctx.cfg = (L, P, fft_length, center, eps, M, n_iter)
        ctx.mode = mode
        ctx.zmean = bool(zmean)
        ctx.floor_db = relative_floor_db
        ctx.images = images
        ctx.with_rt = with_rt
        return mc

    @staticmethod
    @once_differentiable
    def backward(ctx, gmc):
        xc, wc, twiddle, X, hist, G, D, E, av = ctx.saved_tensors
        L, P, fft_length, center, eps, M, n_iter = ctx.cfg
        gmc = gmc.contiguous()
        K = fft_length // 2 + 1
        F = X.numel() // K
        T = xc.size(-1)
        B = xc.numel() // T
        gX = torch.empty_like(X)
        gx = torch.empty_like(xc)
        scratch = torch.empty(_lib.MCEP_BWD_WORKSPACE_BYTES, dtype=torch.uint8, device=gmc.device)
        with torch.cuda.device(gmc.device):
            _call("dsa_mcep_bwd", _p(gmc), _p(X), _p(hist), F, fft_length, M, n_iter, _p(G), _p(D), _p(E), _p(av),
                  _dtype_code(X), _lib.ALGO_AUTO | _lib.ALGO_SCRATCH_HAS_WORKSPACE | (_lib.ALGO_HIST_HAS_RT if ctx.with_rt else 0),
                  _p(ctx.images), _p(scratch), _p(gX), _stream())
            _call("dsa_stft_bwd", _p(gX), _p(xc), B, T, L, P, fft_length, _p(wc), _p(twiddle), int(center), int(ctx.zmean),
                  pad_mode_code(ctx.mode), float(eps), int(ctx.floor_db is not None), 0.0 if ctx.floor_db is None else float(ctx.floor_db), 3,
                  _dtype_code(xc), _lib.ALGO_AUTO, _p(gx), None, _stream())
        return (gx,) + (None,) * 16


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
        images = mcep_images(G, D, E, fft_length, M) if algo != _lib.ALGO_GENERIC else None
        if images is None and not need_hist and algo != _lib.ALGO_GENERIC and _mcep_composed_applies(Xc, M):
            return _mcep_composed_fwd(Xc, G, D, E, av, M, n_iter)
        # the tuned kernels keep every step's rt row next to the iterates (DSA_ALGO_HIST_HAS_RT): the backward skips a chain
        with_rt = need_hist and images is not None and _keep_rt_rows(n_iter, F, M, Xc)
        hist = _mcep_history(n_iter, F, M, X, with_rt) if need_hist else None
        # the tile queue's counters: a per-(device, stream) scratch that the kernel leaves zeroed (no fill launch per call)
        scratch = None
        flag = 0
        if images is not None:
            scratch, flag = _mcep_scratch(X.device)
        if with_rt:
            flag |= _lib.ALGO_HIST_HAS_RT
        with torch.cuda.device(X.device):
            _call("dsa_mcep_fwd", _p(Xc), F, fft_length, M, n_iter, _p(G), _p(D), _p(E), _p(av),
                  _dtype_code(Xc), algo | flag, _p(images), _p(scratch), _p(mc), _p(hist), _stream())
        if need_hist:
            ctx.save_for_backward(Xc, hist, G, D, E, av)
        ctx.cfg = (fft_length, M, n_iter, algo)
        ctx.images = images
        ctx.with_rt = with_rt
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
        images = ctx.images
        # the tuned kernel's scratch with the hand-over area of its split tail (DSA_ALGO_SCRATCH_HAS_WORKSPACE): 1 MB from the
        # caching allocator, stream-ordered
        scratch, flag = None, 0
        if images is not None:
            scratch = torch.empty(_lib.MCEP_BWD_WORKSPACE_BYTES, dtype=torch.uint8, device=gmc.device)
            flag = _lib.ALGO_SCRATCH_HAS_WORKSPACE | (_lib.ALGO_HIST_HAS_RT if ctx.with_rt else 0)
        with torch.cuda.device(gmc.device):
            _call("dsa_mcep_bwd", _p(gmc), _p(Xc), _p(hist), F, fft_length, M, n_iter, _p(G), _p(D), _p(E),
                  _p(av), _dtype_code(Xc), algo | flag, _p(images), _p(scratch), _p(gX), _stream())
        return (gX,) + (None,) * 8


# ----------------------------------------------------------------------------------- mgcep (8(f) row 3)
def gc2gc_fused(c1, out_order, in_gamma, out_gamma, n_fft, twiddle, flags=0):
    """GeneralizedCepstrumToGeneralizedCepstrum._forward (mgc2mgc.py:333-361) in one launch (dsa_gc2gc_fwd): c1:(..., M1+1)
    -> (..., M2+1); forward only.  `flags` folds the scalar steps around it (1 gnorm before, 2 ignorm after, 4 tail * out_gamma,
    8 zeroth * out_gamma + 1).  None when the configuration has no fused kernel (n_fft not a power of two / too long)."""
    _require_device(c1, twiddle)
    _same_dtype(c1, twiddle)
    esz = 8 if c1.dtype == torch.float32 else 16
    if n_fft < 4 or n_fft & (n_fft - 1) or n_fft * esz > 150 * 1024 or out_order + 1 > n_fft:
        return None
    cc = c1.contiguous()
    n_in = cc.size(-1)
    F = cc.numel() // n_in
    out = torch.empty(*cc.shape[:-1], out_order + 1, device=c1.device, dtype=c1.dtype)
    with torch.cuda.device(c1.device):
        _call("dsa_gc2gc_fwd", _p(cc), F, n_in, out_order, float(in_gamma), float(out_gamma), n_fft, _p(twiddle), int(flags),
              _dtype_code(cc), _p(out), _stream())
    return out


class Gc2gcFn(torch.autograd.Function):
    """GeneralizedCepstrumToGeneralizedCepstrum._forward (mgc2mgc.py:333-361) with a graph: dsa_gc2gc_fwd forward, dsa_gc2gc_bwd
    backward -- one launch each, the n_fft-point spectra never in memory.  Use gc2gc_fn(): None when there is no fused kernel."""

    @staticmethod
    def forward(ctx, c1, out_order, in_gamma, out_gamma, n_fft, twiddle):
        y = gc2gc_fused(c1, out_order, in_gamma, out_gamma, n_fft, twiddle)
        ctx.save_for_backward(c1, twiddle)
        ctx.cfg = (out_order, float(in_gamma), float(out_gamma), n_fft)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g2):
        c1, tw = ctx.saved_tensors
        out_order, ig, og, n_fft = ctx.cfg
        cc, gc = c1.contiguous(), g2.contiguous()
        n_in = cc.size(-1)
        F = cc.numel() // n_in
        gc1 = torch.empty_like(cc)
        with torch.cuda.device(cc.device):
            _call("dsa_gc2gc_bwd", _p(cc), _p(gc), F, n_in, out_order, ig, og, n_fft, _p(tw), _dtype_code(cc), _p(gc1), _stream())
        return gc1, None, None, None, None, None


def gc2gc_fn(c1, out_order, in_gamma, out_gamma, n_fft, twiddle):
    """Gc2gcFn.apply where the fused kernels cover the configuration (n_fft a power of two whose five half-length arrays fit LDS),
    else None."""
    esz = 10 if c1.dtype == torch.float32 else 20
    if n_fft < 4 or n_fft & (n_fft - 1) or n_fft * esz + 64 > 150 * 1024 or out_order + 1 > n_fft or not c1.is_cuda:
        return None
    return Gc2gcFn.apply(c1, out_order, in_gamma, out_gamma, n_fft, twiddle)


def mgcep_step(x, b1, images, gamma):
    """(pt, qt, r) of one Newton step of mgcep.py:199-220 in one launch (dsa_mgcep_step: spectrum arithmetic + the five row
    products, float32 / fft_length 512 / cep_order <= 24); forward only."""
    _require_device(x, b1, images)
    _same_dtype(x, b1, images)
    xc, bc = x.contiguous(), b1.contiguous()
    K, M = xc.size(-1), bc.size(-1)
    F = xc.numel() // K
    lead = xc.shape[:-1]
    pt = torch.empty(*lead, M, device=x.device, dtype=x.dtype)
    qt = torch.empty(*lead, 2 * M - 1, device=x.device, dtype=x.dtype)
    r = torch.empty(*lead, M + 1, device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        _call("dsa_mgcep_step", _p(xc), _p(bc), F, 2 * (K - 1), M, float(gamma), _p(images), _dtype_code(xc), _p(pt), _p(qt), _p(r),
              _stream())
    return pt, qt, r


def _step_bwd_entry(images_bwd):
    """The step's adjoint: binary16 images (tables.mgcep_step_bwd_h_images, kept as int16 bit patterns so that Module.float() cannot
    cast them; float16 accepted too) select the binary16 kernel, float32 ones the round-3 kernel.  Anything else is a cast image."""
    if images_bwd.dtype in (torch.int16, torch.float16):
        if images_bwd.numel() != 9 * 22528:
            raise ValueError("mgcep step adjoint: the binary16 operand images must have 9 x 22528 entries")
        return "dsa_mgcep_step_bwd_h"
    if images_bwd.dtype != torch.float32:
        raise ValueError(f"mgcep step adjoint: operand images of dtype {images_bwd.dtype} (expected int16 / float16 or float32)")
    return "dsa_mgcep_step_bwd"


class MgcepStepFn(torch.autograd.Function):
    """(pt, qt, r) of one Newton step of mgcep.py:199-220 with a graph: forward dsa_mgcep_step, backward dsa_mgcep_step_bwd (one
    launch each; float32 / fft_length 512 / cep_order <= 24).  x:(..., 257), b1:(..., M)."""

    @staticmethod
    def forward(ctx, x, b1, images, images_bwd, gamma):
        pt, qt, r = mgcep_step(x, b1, images, gamma)
        ctx.save_for_backward(x, b1, images_bwd)
        ctx.gamma = float(gamma)
        return pt, qt, r

    @staticmethod
    @once_differentiable
    def backward(ctx, gpt, gqt, gr):
        x, b1, images_bwd = ctx.saved_tensors
        xc, bc = x.contiguous(), b1.contiguous()
        K, M = xc.size(-1), bc.size(-1)
        F = xc.numel() // K
        lead = xc.shape[:-1]

        def cot(g, n):
            return torch.zeros(*lead, n, device=xc.device, dtype=xc.dtype) if g is None else g.contiguous()

        gpt, gqt, gr = cot(gpt, M), cot(gqt, 2 * M - 1), cot(gr, M + 1)
        gx = torch.empty_like(xc)
        gb = torch.empty_like(bc)
        with torch.cuda.device(xc.device):
            _call(_step_bwd_entry(images_bwd), _p(xc), _p(bc), _p(gpt), _p(gqt), _p(gr), F, 2 * (K - 1), M, ctx.gamma, _p(images_bwd),
                  _dtype_code(xc), None, _p(gx), _p(gb), _stream())
        return gx, gb, None, None, None


def thsolve_update(pt, qt, r, b1):
    """b1 + solve(symmetric_toeplitz(pt) + hankel(qt), r[..., 1:])  (mgcep.py:226-230) in one call, the right-hand side read
    in place from the step's (.., M + 1) vector (dsa_thsolve_update_fwd: order 24, float32); forward only.  None: not covered."""
    M = pt.size(-1)
    if M != 24 or pt.dtype != torch.float32 or r.size(-1) != M + 1 or not (pt.is_contiguous() and qt.is_contiguous() and r.is_contiguous()):
        return None
    _require_device(pt, qt, r, b1)
    _same_dtype(pt, qt, r, b1)
    lead = tuple(pt.shape[:-1])
    if tuple(qt.shape) != lead + (2 * M - 1,) or tuple(r.shape) != lead + (M + 1,) or tuple(b1.shape) != lead + (M,):
        raise ValueError(f"thsolve_update: shapes {tuple(pt.shape)}, {tuple(qt.shape)}, {tuple(r.shape)}, {tuple(b1.shape)} do not "
                         "describe one batch of order-M systems")
    bc = b1.contiguous()
    F = pt.numel() // M
    out = torch.empty_like(bc)
    with torch.cuda.device(pt.device):
        _call("dsa_thsolve_update_fwd", _p(pt), _p(qt), _p(r), M + 1, 1, F, M, _dtype_code(pt), _p(bc), _p(out), _stream())
    return out


def mgcep_step_solve(x, b1, images_h, gamma, out=None, n_steps=1, want_prev=False):
    """(b1 + solve(toeplitz(pt) + hankel(qt), r[1:]), r) of one WHOLE Newton step of mgcep.py:199-230 in one launch
    (dsa_mgcep_step_solve: binary16-split matrix chains + the block elimination; float32 / fft_length 512 / cep_order 24 /
    gamma in (-1, 0)); forward only.  `images_h`: tables.mgcep_step_h_buffer as a byte tensor; `out`: where the updated coefficients
    go (may be `b1` itself when that is contiguous); `n_steps` Newton steps in the one launch (r is the last step's); `want_prev`: also
    return the last step's input coefficients (what the gain of mgcep.py:221 multiplies r with)."""
    _require_device(x, b1, images_h)
    _same_dtype(x, b1)
    xc, bc = x.contiguous(), b1.contiguous()
    K, M = xc.size(-1), bc.size(-1)
    F = xc.numel() // K
    lead = xc.shape[:-1]
    if out is None:
        out = torch.empty_like(bc)
    elif not (out.is_contiguous() and out.shape == bc.shape and out.dtype == bc.dtype and out.device == bc.device):
        raise ValueError("mgcep_step_solve: `out` must be a contiguous tensor like b1")
    r = torch.empty(*lead, M + 1, device=x.device, dtype=x.dtype)
    prev = torch.empty_like(bc) if want_prev else None
    with torch.cuda.device(x.device):
        _call("dsa_mgcep_step_solve", _p(xc), _p(bc), F, 2 * (K - 1), M, float(gamma), _p(images_h), _dtype_code(xc), _p(out), _p(r), None, None,
              int(n_steps), _p(prev), _stream())
    return (out, r, prev) if want_prev else (out, r)


class MgcepStepSolveFn(torch.autograd.Function):
    """(b1 + solve(toeplitz(pt) + hankel(qt), r[1:]), r) of one Newton step of mgcep.py:199-230 with a graph: forward ONE launch
    (dsa_mgcep_step_solve, which also leaves pt and qt behind), backward the adjoint solve (dsa_thsolve_bwd on the kept system and
    the step's solution) followed by the step's adjoint (dsa_mgcep_step_bwd) -- what autograd composes from MgcepStepFn, ThSolveFn
    and the additions around them, without their intermediate tensors.  float32 / fft_length 512 / cep_order 24."""

    @staticmethod
    def forward(ctx, x, b1, images_h, images_bwd, gamma):
        _require_device(x, b1, images_h)
        _same_dtype(x, b1)
        xc, bc = x.contiguous(), b1.contiguous()
        K, M = xc.size(-1), bc.size(-1)
        F = xc.numel() // K
        lead = xc.shape[:-1]
        out = torch.empty_like(bc)
        r = torch.empty(*lead, M + 1, device=x.device, dtype=x.dtype)
        pt = torch.empty(*lead, M, device=x.device, dtype=x.dtype)
        qt = torch.empty(*lead, 2 * M - 1, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _call("dsa_mgcep_step_solve", _p(xc), _p(bc), F, 2 * (K - 1), M, float(gamma), _p(images_h), _dtype_code(xc), _p(out), _p(r), _p(pt),
                  _p(qt), 1, None, _stream())
        ctx.save_for_backward(xc, bc, out, pt, qt, images_bwd)
        ctx.gamma = float(gamma)
        return out, r

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, gr):
        xc, bc, out, pt, qt, images_bwd = ctx.saved_tensors
        K, M = xc.size(-1), bc.size(-1)
        F = xc.numel() // K
        gout = torch.zeros_like(out) if gout is None else gout.contiguous()
        sol = out - bc
        gp, gq, grhs = torch.empty_like(pt), torch.empty_like(qt), torch.empty_like(sol)
        grf = torch.zeros(*xc.shape[:-1], M + 1, device=xc.device, dtype=xc.dtype) if gr is None else gr.contiguous().clone()
        gx, gb1 = torch.empty_like(xc), torch.empty_like(bc)
        with torch.cuda.device(xc.device):
            _call("dsa_thsolve_bwd", _p(gout), _p(pt), _p(qt), _p(sol), F, M, _dtype_code(pt), _p(gp), _p(gq), _p(grhs), _stream())
            grf[..., 1:] += grhs                                         # the right-hand side is r[1:]
            _call(_step_bwd_entry(images_bwd), _p(xc), _p(bc), _p(gp), _p(gq), _p(grf), F, 2 * (K - 1), M, ctx.gamma, _p(images_bwd), _dtype_code(xc),
                  None, _p(gx), _p(gb1), _stream())
        return gx, gb1 + gout, None, None, None


def mgcep_spectra(x, b1, Cr, Ci, gamma):
    """(5, ..., K): pp, qq (X^2 - Y^2), qq 2XY, pp X, pp Y of one Newton step of mgcep.py:199-209 in one launch
    (dsa_mgcep_spectra); forward only."""
    _require_device(x, b1, Cr, Ci)
    _same_dtype(x, b1, Cr, Ci)
    xc, bc = x.contiguous(), b1.contiguous()
    K, M = xc.size(-1), bc.size(-1)
    F = xc.numel() // K
    out = torch.empty((5, *xc.shape), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        _call("dsa_mgcep_spectra", _p(xc), _p(bc), F, 2 * (K - 1), M, _p(Cr.contiguous()), _p(Ci.contiguous()), float(gamma),
              _dtype_code(xc), _p(out), _stream())
    return out


class ThSolveFn(torch.autograd.Function):
    """g = solve(symmetric_toeplitz(p) + hankel(q), r) per row (mgcep.py:226-229): p:(..., n), q:(..., 2n-1), r:(..., n)."""

    @staticmethod
    def forward(ctx, p, q, r):
        _require_device(p, q, r)
        _same_dtype(p, q, r)
        pc, qc, rc = p.contiguous(), q.contiguous(), r.contiguous()
        n = pc.size(-1)
        if qc.size(-1) != 2 * n - 1 or rc.size(-1) != n:
            raise ValueError("thsolve: expected p:(..., n), q:(..., 2n-1), r:(..., n)")
        if qc.shape[:-1] != pc.shape[:-1] or rc.shape[:-1] != pc.shape[:-1]:   # the kernels index q and r by p's row number
            raise ValueError(f"thsolve: leading dimensions differ (p {tuple(pc.shape)}, q {tuple(qc.shape)}, r {tuple(rc.shape)})")
        F = pc.numel() // n
        g = torch.empty_like(rc)
        with torch.cuda.device(p.device):
            _call("dsa_thsolve_fwd", _p(pc), _p(qc), _p(rc), F, n, _dtype_code(pc), _p(g), _stream())
        ctx.save_for_backward(pc, qc, g)
        return g

    @staticmethod
    @once_differentiable
    def backward(ctx, gg):
        pc, qc, g = ctx.saved_tensors
        n = pc.size(-1)
        F = pc.numel() // n
        ggc = gg.contiguous()
        gp, gq, gr = torch.empty_like(pc), torch.empty_like(qc), torch.empty_like(ggc)
        with torch.cuda.device(gg.device):
            _call("dsa_thsolve_bwd", _p(ggc), _p(pc), _p(qc), _p(g), F, n, _dtype_code(pc), _p(gp), _p(gq), _p(gr), _stream())
        return gp, gq, gr


class ZerodfFn(torch.autograd.Function):
    """Time-variant all-zero filter (zerodf.py:207-243): x:(..., T), b:(..., T/P, M+1) -> y:(..., T)."""

    @staticmethod
    def forward(ctx, x, b, P, zeroth_index, ignore_gain):
        _require_device(x, b)
        _same_dtype(x, b)
        xc, bc = x.contiguous(), b.contiguous()
        T = xc.size(-1)
        M = bc.size(-1) - 1
        B = xc.numel() // max(T, 1)
        # the kernels index the coefficient rows by (utterance, frame): the leading dimensions must agree (zerodf() below
        # broadcasts them the way the reference's tensor arithmetic does before it gets here)
        if bc.dim() < 2 or bc.shape[:-2] != xc.shape[:-1] or bc.size(-2) * P != T:
            raise ValueError(f"zerodf: coefficients {tuple(bc.shape)} do not match the signal {tuple(xc.shape)} at frame period {P}")
        y = torch.empty_like(xc)
        with torch.cuda.device(x.device):
            _call("dsa_zerodf_fwd", _p(xc), _p(bc), B, T, M, P, zeroth_index, int(bool(ignore_gain)), _dtype_code(xc), _p(y), _stream())
        ctx.save_for_backward(xc, bc, y)
        ctx.cfg = (P, zeroth_index, int(bool(ignore_gain)))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xc, bc, y = ctx.saved_tensors
        P, z0, ig = ctx.cfg
        T = xc.size(-1)
        M = bc.size(-1) - 1
        B = xc.numel() // max(T, 1)
        gyc = gy.contiguous()
        gx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(bc) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(gy.device):
            _call("dsa_zerodf_bwd", _p(gyc), _p(xc), _p(bc), _p(y), B, T, M, P, z0, ig, _dtype_code(xc), _p(gx), _p(gb), _stream())
        return gx, gb, None, None, None


def zerodf(x, b, P, zeroth_index, ignore_gain):
    """ZerodfFn with the leading dimensions of x:(..., T) and b:(..., T/P, M+1) broadcast against each other first, as the
    reference's tensor arithmetic does (zerodf.py:207-243 accepts e.g. a batch of signals with one unbatched coefficient
    matrix).  expand() is an autograd operation, so the gradient of a broadcast operand is summed back by autograd."""
    if b.dim() < 2:
        raise ValueError("zerodf: b must have at least two dimensions (frames, coefficients).")
    try:
        batch = torch.broadcast_shapes(x.shape[:-1], b.shape[:-2])
    except RuntimeError as e:
        raise ValueError(f"zerodf: leading dimensions of x {tuple(x.shape)} and b {tuple(b.shape)} do not broadcast") from e
    if tuple(x.shape[:-1]) != tuple(batch):
        x = x.expand(*batch, x.size(-1))
    if tuple(b.shape[:-2]) != tuple(batch):
        b = b.expand(*batch, *b.shape[-2:])
    return ZerodfFn.apply(x, b, P, zeroth_index, ignore_gain)


def zerodf_taylor_shapes_ok(x, b, P) -> bool:
    """Shapes the fused Taylor-stage launches cover, forward and backward (csrc/mgc.hip:zerodf_rows_plan, zerodf_launch_bwd)."""
    return (P % 4 == 0 and 16 <= P <= 256 and b.size(-1) - 1 >= 16 and b.dim() >= 2 and tuple(b.shape[:-2]) == tuple(x.shape[:-1])
            and b.size(-2) * P == x.size(-1) and x.is_cuda and x.dtype == b.dtype and x.dtype in (torch.float32, torch.float64))


def zerodf_taylor_supported(x, b, P) -> bool:
    """zerodf_taylor_shapes_ok and no graph is being recorded."""
    return zerodf_taylor_shapes_ok(x, b, P) and not (torch.is_grad_enabled() and (x.requires_grad or b.requires_grad))


class ZerodfTaylorFn(torch.autograd.Function):
    """y = sum_{i=0}^{order} F^i x / i!  (mglsadf.py:356-365) with a graph: one launch per stage forward (filter, 1 / i, running
    sum: dsa_zerodf_taylor_fwd), one call per stage backward (dsa_zerodf_taylor_bwd: G_{i-1} = gy + F^T G_i / i and
    gb += dF(x_{i-1})^T G_i / i) -- instead of the differentiable filter + two element-wise operations per stage and autograd's
    accumulations.  x:(..., T), b:(..., T/P, M+1), shapes as zerodf_taylor_shapes_ok."""

    @staticmethod
    def forward(ctx, x, b, P, zeroth_index, order):
        xc, bc = x.contiguous(), b.contiguous()
        y = xc.clone()
        cur = xc
        stages = [xc]
        for i in range(1, order + 1):
            cur, y = zerodf_taylor(cur, bc, P, zeroth_index, 1.0 / i, y, want_y=i < order)
            if i < order:
                stages.append(cur)
        ctx.save_for_backward(bc, *stages)
        ctx.cfg = (P, zeroth_index, order)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        bc, *stages = ctx.saved_tensors
        P, z0, order = ctx.cfg
        gy = gy.contiguous()
        T = gy.size(-1)
        M = bc.size(-1) - 1
        B = gy.numel() // max(T, 1)
        gb = torch.zeros_like(bc) if ctx.needs_input_grad[1] else None
        G = gy
        with torch.cuda.device(gy.device):
            for i in range(order, 0, -1):
                G_out = torch.empty_like(gy)
                _call("dsa_zerodf_taylor_bwd", _p(G), _p(stages[i - 1]), _p(bc), B, T, M, P, z0, 1.0 / i, _p(gy), _dtype_code(gy),
                      _p(G_out), _p(gb), _stream())
                G = G_out
        return (G if ctx.needs_input_grad[0] else None), gb, None, None, None


def zerodf_taylor(x, b, P, zeroth_index, scale, acc, want_y=True):
    """One Taylor stage of the multi-stage MLSA filter without a graph (mglsadf.py:356-365): returns
    (scale * zerodf(x; b) or None, acc + scale * zerodf(x; b)) from one launch; `acc` is updated in place."""
    _require_device(x, b, acc)
    _same_dtype(x, b)
    _same_dtype(x, acc)
    xc, bc = x.contiguous(), b.contiguous()
    if not acc.is_contiguous() or acc.shape != xc.shape:
        raise ValueError("zerodf_taylor: acc must be a contiguous tensor of the signal's shape")
    if bc.dim() < 2 or bc.shape[:-2] != xc.shape[:-1] or bc.size(-2) * P != xc.size(-1):
        raise ValueError(f"zerodf_taylor: coefficients {tuple(bc.shape)} do not match the signal {tuple(xc.shape)} at frame period {P}")
    T = xc.size(-1)
    M = bc.size(-1) - 1
    B = xc.numel() // max(T, 1)
    y = torch.empty_like(xc) if want_y else None
    with torch.cuda.device(x.device):
        _call("dsa_zerodf_taylor_fwd", _p(xc), _p(bc), B, T, M, P, zeroth_index, float(scale), _p(acc), _dtype_code(xc),
              _p(y), _p(acc), _stream())
    return y, acc


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
        scratch = _scratch(x.device)
        with torch.cuda.device(x.device):
            _call("dsa_lpc_fwd", _p(xc), F, L, M, float(eps), _dtype_code(xc), _p(scratch), _p(out), _stream())
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


def frame_window_lpc(x, window, L, P, M, eps, center=True, mode="constant", exact_lag_sums=False):
    """Fused LPC branch, forward: LPC(Window(Frame(x))), README.md:198-201 of the reference.  exact_lag_sums: float64 lag sums on
    the vector unit instead of binary16 splits on the matrix pipe (DSA_LPC_EXACT_LAGSUMS; for near-singular frames)."""
    _require_device(x, window)
    _same_dtype(x, window)
    xc, wc = x.contiguous(), window.contiguous()
    T = xc.size(-1)
    B = xc.numel() // T
    out = torch.empty(*xc.shape[:-1], num_frames(T, P), M + 1, device=x.device, dtype=x.dtype)
    # the kept-zero per-(device, stream) counters of the persistent kernels (the kernel hands them back zeroed: no fill launch per call;
    # under graph capture a private block and the library's own reset, see _mcep_scratch)
    scratch, flag = _mcep_scratch(x.device)
    if exact_lag_sums:
        flag |= _lib.LPC_EXACT_LAGSUMS
    with torch.cuda.device(x.device):
        _call("dsa_frame_window_lpc_fwd", _p(xc), B, T, L, P, _p(wc), int(center), pad_mode_code(mode) | flag, M,
              float(eps), _dtype_code(xc), _p(scratch), _p(out), _stream())
    return out


def frame_window_lpc_bwd_supported(x, L, P, M, center, mode) -> bool:
    """dsa_frame_window_lpc_bwd takes this configuration (the conditions of its launcher, csrc/lpc.hip): float32, lpc_order 24,
    25 <= frame_length <= 512, constant padding, and a (frame_length, frame_period) pair whose overlap fits the wave's stretch."""
    if not (x.is_cuda and x.dtype == torch.float32 and M == 24 and 25 <= L <= 512 and mode == "constant" and P >= 1 and x.size(-1) >= 1):
        return False
    left = L // 2 if center else 0
    halo = (left - 1) // P - (left - L) // P          # (Python's floor division, as lb_floordiv)
    return L + P <= 1024 and halo <= 48 and min(64 - halo, num_frames(x.size(-1), P)) >= 1


class FrameWindowLpcFn(torch.autograd.Function):
    """LPC(Window(Frame(x))) (README.md:198-201 of the reference; frame.py:120-141, window.py:185-193, lpc.py:137-139) as ONE launch
    forward (dsa_frame_window_lpc_fwd) and ONE launch backward (dsa_frame_window_lpc_bwd): no (B N, L) tensor in memory either
    way.  A fixed window; callers check frame_window_lpc_bwd_supported first."""

    @staticmethod
    def forward(ctx, x, window, L, P, M, eps, center, exact_lag_sums):
        out = frame_window_lpc(x, window, L, P, M, eps, center, "constant", exact_lag_sums)
        ctx.save_for_backward(x.contiguous(), window.contiguous())
        ctx.cfg = (L, P, M, eps, center)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xc, wc = ctx.saved_tensors
        L, P, M, eps, center = ctx.cfg
        gc = g.contiguous()
        T = xc.size(-1)
        gx = torch.empty_like(xc)
        with torch.cuda.device(g.device):
            _call("dsa_frame_window_lpc_bwd", _p(gc), _p(xc), xc.numel() // T, T, L, P, _p(wc), int(center), pad_mode_code("constant"),
                  M, float(eps), _dtype_code(xc), _p(gx), _stream())
        return gx, None, None, None, None, None, None, None
