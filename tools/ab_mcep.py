"""A/B of the tuned mcep forward variants (dev tool): DSA_MCEP_VARIANT=<v> python tools/ab_mcep.py [B].
Prints kernel time at B utterances (200 frames each) and the deviation from the float64 generic path
on the first 4096 frames."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 16000, generator=g).to(dev)
stft = dsp.STFT(400, 80, 512).to(dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10).to(dev)
X = stft(x)
y = mcep(X)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts = []
for _ in range(20):
    ev[0].record()
    y = mcep(X)
    ev[1].record()
    torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
ts.sort()
Xs = X.reshape(-1, 257)[:4096]
ref = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10).to(dev).double()(Xs.double())
err = (y.reshape(-1, 25)[:4096].double() - ref).abs()
tol = 2e-5 + 1e-4 * ref.abs()
print(f"variant {os.environ.get('DSA_MCEP_VARIANT', 'default')}: median {ts[len(ts)//2]:.4f} ms  min {ts[0]:.4f} ms  "
      f"frames {X.shape[0] * X.shape[1]}  max|err| {err.max().item():.3e}  max err/tol {(err / tol).max().item():.4f}  "
      f"nan {torch.isnan(y).any().item()}")
