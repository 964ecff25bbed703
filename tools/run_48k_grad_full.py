"""STFT -> mel-cepstrum at a 48 kHz set-up WITH a gradient back to the waveform (B utterances x 1 s): for kernel traces / timing.
usage: python tools/run_48k_grad_full.py [B] [2048|1024]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nfft = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
fl, fp, M = (1200, 240, 49) if nfft == 2048 else (800, 200, 34)
x = torch.randn(B, 48000, generator=torch.Generator().manual_seed(0)).to(dev)
stft = dsp.STFT(fl, fp, nfft, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
def step():
    xg = x.clone().requires_grad_(True)
    mcep(stft(xg)).sum().backward()
    return xg.grad
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("N", "5"))
for _ in range(n): step()
torch.cuda.synchronize()
print(f"B={B} nfft={nfft}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per forward + backward ({B * (48000 // fp + 1)} frames)")
