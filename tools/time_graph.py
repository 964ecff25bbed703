"""Launch-bound calls eager against one HIP-graph replay (diffsptk_amd.Graphed): small batches of 1 s utterances, float32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=10, device=dev)
ml = dsp.MLSA(24, 80, alpha=0.42, mode="multi-stage", device=dev)
def wall(fn, n=20):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for B in (1, 8, 64):
    x = torch.randn(B, 16000, device=dev)
    cases = {"STFT -> mcep": lambda x: mcep(stft(x)),
             "STFT -> mgcep (gamma -0.5)": lambda x: mg(stft(x)),
             "STFT -> mcep -> MLSA multi-stage": lambda x: ml(x, mcep(stft(x))[:, :200])}
    for name, fn in cases.items():
        g = dsp.Graphed(fn, x)
        print(f"batch {B:3d}  {name:34s} eager {wall(lambda: fn(x)):7.3f} ms   graph replay {wall(lambda: g(x)):7.3f} ms")
