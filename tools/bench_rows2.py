"""Timings of the SURVEY 8(f) rows 3-4 modules (cepstral analysis with refinement, mgcep, conversions, MLSA filter).
Usage: python tools/bench_rows2.py [utterances=256]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
x = torch.randn(B, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
frames = B * 200


def gpu_time(fn, reps=10, ramp=0.2):
    fn()
    t0 = time.time()
    while time.time() - t0 < ramp:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def row(name, fn, reps=10):
    t = gpu_time(fn, reps)
    print(f"{name:58s} {t:9.3f} ms  {frames / t / 1e3:9.1f} Mframes/s", flush=True)


with torch.no_grad():
    X = stft(x)
    print(f"{B} utterances x 1 s = {frames} frames")
    for it in (0, 1, 3):
        ca = dsp.CepstralAnalysis(fft_length=512, cep_order=24, n_iter=it, device=dev)
        row(f"CepstralAnalysis n_iter={it}", lambda: ca(X))
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
    row("MelCepstralAnalysis n_iter=10 (gamma = 0)", lambda: mcep(X))
    for gamma, it in ((-0.5, 10), (-1.0, 10), (-0.5, 2)):
        mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=gamma, n_iter=it, device=dev)
        row(f"MelGeneralizedCepstralAnalysis gamma={gamma} n_iter={it}", lambda: mg(X), reps=3)
    mc = mcep(X)
    m2b = dsp.MelCepstrumToMLSADigitalFilterCoefficients(24, 0.42, device=dev)
    row("mc2b", lambda: m2b(mc))
    g2g = dsp.MelGeneralizedCepstrumToMelGeneralizedCepstrum(24, 30, in_alpha=0.42, out_alpha=0.0, in_gamma=0.0, out_gamma=-0.5, device=dev)
    row("mgc2mgc 24 -> 30 (alpha 0.42 -> 0, gamma 0 -> -0.5)", lambda: g2g(mc))
    m2s = dsp.MelGeneralizedCepstrumToSpectrum(24, 512, alpha=0.42, device=dev)
    row("mgc2sp (power)", lambda: m2s(mc))
    e = torch.randn(B, 16000, device=dev)
    for mode, kw in (("multi-stage", {}), ("single-stage", {}), ("freq-domain", dict(frame_length=400, fft_length=512))):
        ml = dsp.MLSA(24, 80, alpha=0.42, mode=mode, device=dev, **kw)
        row(f"MLSA {mode}", lambda: ml(e, mc), reps=3)
for B2 in (256,):
    xg = x[:B2].clone().requires_grad_(True)
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=10, device=dev)

    def fb():
        xg.grad = None
        mg(stft(xg)).mean().backward()

    row("STFT + mgcep gamma=-0.5 forward + backward", fb, reps=2)
