"""Lane-level numpy model of the filter-bank epilogue of the fused STFT -> mel filter bank kernel
(csrc/stft_pk.h, FB variant): the per-lane plan `tables.fbank_scan_plan` builds from a filter-bank matrix whose rows
have at most two adjacent non-zero channels, and the segmented DPP scan network that sums each interval between
channel centres over the lanes that hold its bins.  Run: python tools/proto_fbank_scan.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsptk_amd.utils import tables  # noqa: E402


def dpp_row_shr(x, d):
    out = np.zeros_like(x)            # bound_ctrl:1 -> invalid source lanes read 0
    for lane in range(64):
        if lane % 16 >= d:
            out[lane] = x[lane - d]
    return out


def dpp_row_bcast15(x):               # rows 1..3 read lane 15 of the previous row (row_mask selects the writers)
    out = np.zeros_like(x)
    for lane in range(16, 64):
        out[lane] = x[(lane // 16) * 16 - 1]
    return out


def dpp_row_bcast31(x):               # rows 2, 3 read lane 31
    out = np.zeros_like(x)
    for lane in range(32, 64):
        out[lane] = x[31]
    return out


def dpp_wave_shr1(x):
    out = np.zeros_like(x)
    out[1:] = x[:-1]
    return out


def wave_epilogue(P, plan, C):
    """P: (257,) float32 power (or amplitude) values of one frame -> (C,) channel sums, the way a wave computes them."""
    f32 = np.float32
    lane = np.arange(64)
    sp0 = np.stack([P[2 * lane + 1], P[255 - 2 * lane]], 1).astype(f32)   # (lane, half): element e0
    sp1 = np.stack([P[2 * lane + 2], P[254 - 2 * lane]], 1).astype(f32)   # element e1
    slots = np.zeros((2, 2, 128), f32)          # [q: 0 = down (-> channel j - 1), 1 = up (-> channel j)][half][j]
    for h in range(2):
        for q, (w0, w1) in enumerate(((plan["wd0"], plan["wd1"]), (plan["wu0"], plan["wu1"]))):
            c0 = (sp0[:, h] * w0[:, h]).astype(f32)
            c1 = (sp1[:, h] * w1[:, h]).astype(f32)
            x = (c0 * plan["nb"][:, h] + c1).astype(f32)
            for step, d in enumerate((1, 2, 4, 8)):
                x = (x + dpp_row_shr(x, d) * plan["mask"][:, h, step]).astype(f32)
            t = (x + dpp_row_bcast15(x) * plan["mask"][:, h, 4]).astype(f32)
            rows13 = (lane // 16) % 2 == 1
            x = np.where(rows13, t, x)
            t = (x + dpp_row_bcast31(x) * plan["mask"][:, h, 5]).astype(f32)
            x = np.where(lane >= 32, t, x)
            m = (c0 + dpp_wave_shr1(x) * plan["mM"][:, h]).astype(f32)
            for ln in range(64):
                if plan["isE"][ln, h]:
                    slots[q, h, plan["jE"][ln, h]] = x[ln]
                if plan["isM"][ln, h]:
                    slots[q, h, plan["jM"][ln, h]] = m[ln]
    c = np.arange(C)
    y = slots[1, 0, c] + slots[1, 1, c] + slots[0, 0, c + 1] + slots[0, 1, c + 1]
    return (y + plan["h0"][:C] * P[0] + plan["h256"][:C] * P[256]).astype(f32)


def main():
    rng = np.random.default_rng(0)
    for (nfft, C, sr, fmin, fmax) in ((512, 40, 16000, 0, None), (512, 80, 16000, 0, None), (512, 20, 16000, 0, None),
                                       (512, 24, 16000, 64, 7600), (512, 62, 48000, 20, 20000), (512, 3, 8000, 0, None)):
        H = np.asarray(tables.fbank_matrix(nfft, C, sr, fmin, fmax, "htk", None))
        plan = tables.fbank_scan_plan(H)
        assert plan is not None, (C, sr)
        worst = 0.0
        for trial in range(20):
            P = (rng.standard_normal(257) ** 2 * 10.0 ** rng.uniform(-6, 6, 257)).astype(np.float32)
            if trial == 0:
                P[:] = 1.0
            y = wave_epilogue(P, plan, C)
            ref = P.astype(np.float64) @ H
            worst = max(worst, float(np.max(np.abs(y - ref) / np.maximum(np.abs(ref), 1e-300))))
        print(f"C={C} sr={sr} f=[{fmin},{fmax}]: max relative error {worst:.2e}")
        assert worst < 2e-6
    # matrices without the structure are refused
    Hd = np.abs(rng.standard_normal((257, 40)))
    assert tables.fbank_scan_plan(Hd) is None
    He = np.asarray(tables.fbank_matrix(512, 40, 16000, 0, None, "htk", 1.0))
    print("erb filters:", "plan" if tables.fbank_scan_plan(He) is not None else "refused (overlapping filters)")
    print("ok")


if __name__ == "__main__":
    main()
