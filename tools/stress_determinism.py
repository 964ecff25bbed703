"""Both STFT -> mel-cepstrum paths, N launches at the bench size, every output compared bit for bit with the first (guards against the
rare stale-result interaction of DESIGN.md 3.25 at rates far below what the five launches of the test suite would catch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = "cuda"
x = torch.randn(1024, 16000, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
fused = dsp.fuse(stft, mcep)
with torch.no_grad():
    ref2, ref1 = mcep(stft(x)), fused(x)
    bad2 = bad1 = 0
    for i in range(N):
        bad2 += int((mcep(stft(x)) != ref2).any(-1).sum())
        bad1 += int((fused(x) != ref1).any(-1).sum())
print(f"{N} launches x 204800 frames: frames differing from the first launch: two kernels {bad2}, one launch {bad1}; "
      f"max |one launch - two kernels| {float((ref1 - ref2).abs().max()):.3e}")
