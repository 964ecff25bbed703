"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Second restatement of the reference hot path, written with stock PyTorch CPU operators
(the same ATen kernels the reference dispatches to: ``torch.fft.rfft/irfft``, ``matmul``,
``torch.linalg.solve``, ``Tensor.unfold``, ``F.pad``).  It exists for two reasons:

* autograd through these functions gives a gradient oracle on ANY input (the reference has
  no hand-written backward: SURVEY.md section 3.5), usable on the GPU box where the
  reference itself is absent;
* timed on the host cores it is the "reference CPU path" column of ``bench.py``
  (``cpu_baseline.kind = "port"``).

Written from the reference's behaviour (file:line cited per function), not copied from it:
compact pure functions, no module/precompute machinery.  Pinned against the golden
vectors in ``tests/test_oracle_golden.py``.  Never imported by the product package.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _pad_frames(x, L, P, center, mode):
    # frame.py:130-138
    left, right = (L // 2, (L - 1) // 2) if center else (0, L - 1)
    if mode != "constant" and x.dim() == 1:
        xp = F.pad(x[None], (left, right), mode=mode)[0]
    else:
        xp = F.pad(x, (left, right), mode=mode)
    return xp.unfold(-1, L, P)


def frame(x, L, P, center=True, zmean=False, mode="constant"):
    y = _pad_frames(x, L, P, center, mode)
    if zmean:  # frame.py:139-140
        y = y - y.mean(-1, keepdim=True)
    return y


def window_table(L, window="blackman", norm="power", symmetric=True, dtype=torch.float64):
    """window.py:134-183, cosine-sum family + rectangular only (what the baseline uses)."""
    n = torch.arange(L, dtype=torch.float64)
    D = (L - 1) if symmetric else L
    ph = 2 * math.pi * n / max(D, 1)
    if window in (0, "blackman"):
        w = 0.42 - 0.5 * torch.cos(ph) + 0.08 * torch.cos(2 * ph)
    elif window in (1, "hamming"):
        w = 0.54 - 0.46 * torch.cos(ph)
    elif window in (2, "hanning"):
        w = 0.5 - 0.5 * torch.cos(ph)
    elif window in (5, "rectangular"):
        w = torch.ones(L, dtype=torch.float64)
    else:
        raise ValueError("torch_port.window_table: unsupported window")
    if L == 1:
        w = torch.ones(1, dtype=torch.float64)
    if norm in (1, "power"):
        w = w / torch.sqrt((w * w).sum())
    elif norm in (2, "magnitude"):
        w = w / w.sum()
    return w.to(dtype)


def stft_power(x, L=400, P=80, n=512, w=None, eps=1e-9, center=True, zmean=False, mode="constant"):
    """stft.py:237-241 with window.py:185-193, fftr.py:145,117 and spec.py:173."""
    if w is None:
        w = window_table(L, dtype=x.dtype)
    fr = frame(x, L, P, center, zmean, mode) * w
    fr = F.pad(fr, (0, n - L))
    X = torch.fft.rfft(fr, n=n).abs()
    return torch.square(X) + eps


def warp_matrix(in_order, out_order, alpha, dtype, coefficients=False):
    """freqt.py:123-139 (coefficients=False) / mcep.py:272-284 (True); returns (L1, L2)."""
    L1, L2 = in_order + 1, out_order + 1
    A = [[0.0] * L1 for _ in range(L2)]
    if coefficients:
        for i in range(L2):
            A[i][0] = (-alpha) ** i
        first = 1
    else:
        for j in range(L1):
            A[0][j] = alpha ** j
        if L1 > 1 and L2 > 1:
            for j in range(1, L1):
                A[1][j] = A[0][j - 1] * (1 - alpha * alpha) * j
        first = 2
    for i in range(first, L2):
        for j in range(1, L1):
            A[i][j] = A[i - 1][j - 1] + alpha * (A[i][j - 1] - A[i - 1][j])
    return torch.tensor(A, dtype=torch.float64).T.contiguous().to(dtype)


class McepTables:
    def __init__(self, n=512, M=24, alpha=0.42, dtype=torch.float32):
        H = n // 2
        self.n, self.M, self.H = n, M, H
        self.A_f = warp_matrix(H, M, alpha, dtype)            # mcep.py:145-155
        self.A_i = warp_matrix(M, H, -alpha, dtype)           # mcep.py:156-166
        self.A_r = warp_matrix(H, 2 * M, alpha, dtype, True)  # mcep.py:167-177
        self.av = (-alpha) ** torch.arange(M + 1, dtype=dtype)  # mcep.py:179-181


def _toeplitz(r):
    # private.py:291-295: R[i, j] = r[|i-j|]
    d = r.size(-1)
    idx = (torch.arange(d)[:, None] - torch.arange(d)[None, :]).abs()
    return r[..., idx]


def _hankel(rt):
    # private.py:298-302: Q[i, j] = rt[i+j]
    m = (rt.size(-1) + 1) // 2
    idx = torch.arange(m)[:, None] + torch.arange(m)[None, :]
    return rt[..., idx]


def mcep(X, tab: McepTables, n_iter=10):
    """mcep.py:189-224."""
    H, M = tab.H, tab.M
    log_x = torch.log(X)
    c = torch.fft.irfft(log_x)
    scale = torch.ones(c.size(-1), dtype=c.dtype)
    scale[0] = 0.5
    scale[H] = 0.5
    mc = (c * scale)[..., : H + 1] @ tab.A_f
    for _ in range(n_iter):
        d = torch.fft.rfft(mc @ tab.A_i, n=tab.n).real
        e = torch.exp(log_x - d - d)
        rt = torch.fft.irfft(e)[..., : H + 1] @ tab.A_r
        r = rt[..., : M + 1]
        mc = mc + torch.linalg.solve(_toeplitz(r) + _hankel(rt), r - tab.av)
    return mc


def acorr(x, M):
    """acorr.py:110-120 (naive format)."""
    n = x.size(-1) + M
    n += n % 2
    return torch.fft.irfft(torch.fft.rfft(x, n=n).abs().square())[..., : M + 1]


def levdur(r, eps):
    """levdur.py:113-127."""
    M = r.size(-1) - 1
    R = _toeplitz(r[..., :-1]) + eps * torch.eye(M, dtype=r.dtype)
    a = torch.linalg.solve(R, -r[..., 1:, None])[..., 0]
    K = torch.sqrt((r[..., 1:] * a).sum(-1, keepdim=True) + r[..., :1])
    return torch.cat([K, a], -1)


def frame_window_lpc(x, L=400, P=80, M=24, eps=1e-5, w=None):
    """README.md:198-201: LPC(Window(Frame(x)))."""
    if w is None:
        w = window_table(L, dtype=x.dtype)
    return levdur(acorr(frame(x, L, P) * w, M), eps)


def stft_mcep(x, tab: McepTables, L=400, P=80, n_iter=10, w=None):
    return mcep(stft_power(x, L, P, tab.n, w), tab, n_iter)
