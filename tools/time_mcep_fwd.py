"""One-line timing of the tuned mel-cepstral forward for A/B runs of library builds (tools/gpu_ab_lib.sh): median / min of 30
launches at 1024 utterances x 200 frames, max deviation from the float64 generic path on 4096 frames."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = "cuda"
x = torch.randn(1024, 16000, generator=torch.Generator().manual_seed(0)).to(dev)
stft = dsp.STFT(400, 80, 512).to(dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10).to(dev)
with torch.no_grad():
    X = stft(x)
    for _ in range(300):   # clock ramp
        y = mcep(X)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = mcep(X)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    Xs = X.reshape(-1, 257)[:4096]
    ref = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10).to(dev).double()(Xs.double())
err = (y.reshape(-1, 25)[:4096].double() - ref).abs()
print(f"{tag}: mcep fwd median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f} ms  max|err| {err.max().item():.3e}  nan {torch.isnan(y).any().item()}")
