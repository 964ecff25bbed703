"""STFT -> mel filter bank (-> MFCC): the fused launch against the two-stage path, HIP-event timings.
Usage: python tools/bench_fused.py [utterances=1024]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
x = torch.randn(B, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
frames = B * 200


def gpu_time(fn, reps=50, ramp=0.3):
    import time
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


with torch.no_grad():
    print(f"{B} utterances x 1 s = {frames} frames; STFT alone {gpu_time(lambda: stft(x)) * 1e3:.1f} us")
    for name, ana in (("fbank C=40 power", dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, use_power=True, device=dev)),
                      ("fbank C=40 amplitude", dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, device=dev)),
                      ("fbank C=80 power", dsp.MelFilterBankAnalysis(fft_length=512, n_channel=80, sample_rate=16000, use_power=True, device=dev)),
                      ("MFCC 12 / C=40", dsp.MFCC(fft_length=512, mfcc_order=12, n_channel=40, sample_rate=16000, lifter=22, device=dev))):
        f = dsp.fuse(stft, ana)
        y = f(x)
        assert f.last_path == "fused"
        err = (y - ana(stft(x))).abs().max().item()
        tf = gpu_time(lambda: f(x))
        t2 = gpu_time(lambda: ana(stft(x)))
        byts = frames * (320 + 4 * y.size(-1))
        print(f"{name:22s} fused {tf * 1e3:7.1f} us ({frames / tf / 1e3:7.1f} Mframes/s, {byts / tf / 1e6:6.0f} GB/s algorithmic)"
              f"   two-stage {t2 * 1e3:7.1f} us   x{t2 / tf:.2f}   max |diff| {err:.2e}")
