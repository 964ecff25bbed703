"""get_alpha / read (diffsptk/utils/public.py:22-157), config-1 plumbing."""
from __future__ import annotations

import wave

import numpy as np
import torch

_HTS_ALPHA = {8000: 0.31, 10000: 0.35, 12000: 0.37, 16000: 0.42, 22050: 0.45, 24000: 0.47,
              32000: 0.50, 44100: 0.53, 48000: 0.55}


def _auto_alpha(sample_rate: int, n_freq: int, n_alpha: int) -> float:
    """Grid search for the all-pass coefficient whose phase response is closest (L2) to the
    mel scale log(1 + f/1000) (public.py:74-96)."""
    grid = np.arange(n_freq, dtype=np.float64) / (n_freq - 1)      # 0 .. 1 (Nyquist)
    mel = np.log1p(grid * 0.5 * sample_rate / 1000.0)
    mel *= np.pi / mel[-1]
    omega = grid * np.pi
    best, best_err = 0.0, np.inf
    for a in np.linspace(0.0, 1.0, n_alpha, endpoint=False):
        with np.errstate(divide="ignore", invalid="ignore"):
            warped = np.arctan((1 - a * a) * np.sin(omega) / ((1 + a * a) * np.cos(omega) - 2 * a))
        warped = np.where(warped < 0, warped + np.pi, warped)
        err = float(np.square(mel - warped).sum())
        if err < best_err:
            best, best_err = float(a), err
    return best


def get_alpha(sample_rate: int, mode: str = "hts", n_freq: int = 10, n_alpha: int = 100) -> float:
    """Frequency-warping factor for a sample rate (public.py:22-104; 16 kHz -> 0.42)."""
    if mode == "hts":
        sr = int(sample_rate)
        if sr not in _HTS_ALPHA:
            raise ValueError(f"Unsupported sample rate: {sample_rate}. Please use mode='auto'.")
        return _HTS_ALPHA[sr]
    if mode == "auto":
        return _auto_alpha(sample_rate, n_freq, n_alpha)
    raise ValueError("Only hts and auto are supported.")


def read(filename: str, device=None, dtype=None, channel_first: bool = True, **kwargs):
    """Read a PCM wav file into a float tensor in [-1, 1) (stdlib ``wave``; the reference uses
    soundfile, public.py:152-156, which scales int16 by 1/32768 as done here).  Of the keyword arguments the reference hands to
    ``soundfile.read`` the ones that select samples are honoured -- ``start``, ``stop``, ``frames`` (negative values count from the
    end, as there) and ``always_2d``; anything else raises ``TypeError``."""
    start, stop, frames = kwargs.pop("start", 0), kwargs.pop("stop", None), kwargs.pop("frames", -1)
    always_2d = bool(kwargs.pop("always_2d", False))
    if kwargs:
        raise TypeError(f"read() got unsupported soundfile arguments: {sorted(kwargs)}")
    with wave.open(filename, "rb") as w:
        nch, width, sr, total = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        start = max(0, min(total, start + total if start < 0 else start))
        if stop is not None and frames >= 0:
            raise TypeError("Only one of {frames, stop} may be used")
        end = total if stop is None else max(0, min(total, stop + total if stop < 0 else stop))
        if frames >= 0:
            end = min(total, start + frames)
        n = max(0, end - start)
        w.setpos(start) if start < total else None
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width: {width}")
    x = x.reshape(-1, nch)
    x = x[:, 0] if (nch == 1 and not always_2d) else (x.T if channel_first else x)
    if dtype is None:
        dtype = torch.get_default_dtype()
    return torch.tensor(np.ascontiguousarray(x), device=device, dtype=dtype), sr


def write(filename: str, x, sample_rate: int, channel_first: bool = True, **kwargs) -> None:
    """Write a waveform (C, T) / (T, C) / (T,) to a PCM wav file (stdlib ``wave``; the reference hands the array to
    ``soundfile.write``, public.py:160-198, whose default for .wav is 16-bit PCM with the float samples scaled by 32767 and
    rounded to nearest-even -- done the same way here, saturating instead of wrapping).  ``subtype="PCM_16"`` (default) or
    ``"PCM_32"``; other soundfile arguments raise ``TypeError``."""
    subtype = kwargs.pop("subtype", None) or "PCM_16"
    if kwargs:
        raise TypeError(f"write() got unsupported soundfile arguments: {sorted(kwargs)}")
    if subtype not in ("PCM_16", "PCM_32"):
        raise ValueError(f"unsupported subtype: {subtype}")
    a = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    a = a.astype(np.float64)
    if a.ndim == 2:
        a = a.T if channel_first else a            # (T, C): interleaved frames
    elif a.ndim != 1:
        raise ValueError("x must be (T,), (C, T) or (T, C)")
    nch = 1 if a.ndim == 1 else a.shape[1]
    if subtype == "PCM_16":
        pcm = np.clip(np.rint(a * 32767.0), -32768, 32767).astype("<i2")
        width = 2
    else:
        pcm = np.clip(np.rint(a * 2147483647.0), -2147483648, 2147483647).astype("<i4")
        width = 4
    with wave.open(filename, "wb") as w:
        w.setnchannels(nch)
        w.setsampwidth(width)
        w.setframerate(int(sample_rate))
        w.writeframes(np.ascontiguousarray(pcm).tobytes())
