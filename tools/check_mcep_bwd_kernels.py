import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from oracle import torch_port as TP
DEV="cuda"
gen = torch.Generator().manual_seed(33)
stft = dsp.STFT(400, 80, 512, device=DEV)
x = torch.randn(41, 4800, generator=gen).to(DEV)
X = stft(x).reshape(-1, 257)[:2449]
tab = TP.McepTables(512, 24, 0.42, torch.float64)
for n_iter in (1, 3, 10):
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=n_iter, device=DEV)
    w = torch.randn(2449, 25, generator=gen).to(DEV)
    def grad():
        Xg = X.detach().clone().requires_grad_(True)
        (mcep(Xg) * w).sum().backward()
        return Xg.grad
    sel = list(range(0, 2449))
    Xs = X[sel].double().cpu().requires_grad_(True)
    (TP.mcep(Xs, tab, n_iter) * w[sel].double().cpu()).sum().backward()
    ref = Xs.grad
    for v in ("1", "0"):
        os.environ["DSA_MCEP_BWD2"] = v
        g = grad()[sel].double().cpu()
        per = ((g - ref).abs().amax(1) / ref.abs().amax(1))
        print(f"n_iter {n_iter} BWD2={v}: max over frames of (max |err| / max |ref|) {float(per.max()):.3e}  median {float(per.median()):.3e}  global {float((g-ref).abs().max()/ref.abs().max()):.3e}")
    os.environ["DSA_MCEP_HIST_RT"] = "0"
    g = grad()[sel].double().cpu()
    del os.environ["DSA_MCEP_HIST_RT"]
    per = ((g - ref).abs().amax(1) / ref.abs().amax(1))
    print(f"n_iter {n_iter} recompute: {float(per.max()):.3e} median {float(per.median()):.3e} global {float((g-ref).abs().max()/ref.abs().max()):.3e}")
