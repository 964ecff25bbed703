"""Does alternating whole steps between two streams fill the mel-cepstral kernel's tail with the next step's STFT?
(dev tool)  python tools/ab_pipeline.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffsptk_amd as dsp

dev = "cuda"
x = torch.randn(1024, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(nstream, K=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(K):
            if nstream == 1:
                mcep(stft(x))
            else:
                with torch.cuda.stream(streams[i % nstream]):
                    mcep(stft(x))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for _ in range(8):   # clock ramp
    run(1, 40); run(2, 40)
for n in (1, 2, 1, 2):
    print(f"{n} stream(s): {run(n, 200):.4f} ms/step")
