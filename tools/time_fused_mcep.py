#!/usr/bin/env python3
"""STFT -> mcep at the bench size (1024 utterances x 1 s): the one-launch path against the two kernels, back to back and
interleaved (events on the current stream).  usage: python tools/time_fused_mcep.py [B]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
x = torch.randn(B, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
fused = dsp.fuse(stft, mcep)


def timeit(fn, n=50, groups=5):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n)
    return statistics.median(out), min(out), max(out)


with torch.no_grad():
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:   # clock ramp
        fused(x)
    torch.cuda.synchronize()
    X = stft(x)
    for rnd in range(3):
        a = timeit(lambda: mcep(stft(x)))
        b = timeit(lambda: fused(x))
        c = timeit(lambda: mcep(X))
        d = timeit(lambda: stft(x))
        print(f"round {rnd}: two kernels {a[0]:.4f} ms ({a[1]:.4f}-{a[2]:.4f}) | fused {b[0]:.4f} ms ({b[1]:.4f}-{b[2]:.4f}) | mcep alone "
              f"{c[0]:.4f} | stft alone {d[0]:.4f} | frames/s fused {B * 200 / b[0] * 1e3:.4g} two {B * 200 / a[0] * 1e3:.4g}")
    assert fused.last_path == "fused"
