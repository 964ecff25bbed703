"""Prints the MEASURED errors behind the float32 tolerances of the GPU tests (so that the bounds in tests/ can be set to a small
multiple of them): config-3 gradient, mel-generalized analysis on speech and on random spectra, smoke()'s mel-cepstrum."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffsptk_amd as dsp  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import torch_port as TP  # noqa: E402

DEV = "cuda"
# config 3 gradient (tests/test_gpu_configs.py::test_config3_backward_batch256_vs_float64_autograd)
B = 256
x = torch.randn(B, 16000, generator=torch.Generator().manual_seed(3))
xd = x.to(DEV).requires_grad_(True)
stft = dsp.STFT(400, 80, 512, device=DEV)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=DEV)
mc = mcep(stft(xd))
mc.mean().backward()
g = xd.grad.cpu().numpy()
sel = [0, 37, 101, 128, 200, 255]
tab = TP.McepTables(512, 24, 0.42, torch.float64)
xs = x[sel].double().requires_grad_(True)
mcs = TP.stft_mcep(xs, tab)
(mcs.sum() / mc.numel()).backward()
ref = xs.grad.numpy()
print("config3 grad: max |err| / max |ref| per utterance:", [float(np.abs(g[b] - ref[i]).max() / np.abs(ref[i]).max()) for i, b in enumerate(sel)])
print("config3 mcep: max |err|:", float(np.abs(mc.detach().cpu().numpy()[sel] - mcs.detach().numpy()).max()))
# mgcep on speech (tests/test_gpu_synth.py::test_mgcep_speech_512_and_gamma0_route)
gd = np.load(os.path.join(ROOT, "tests", "golden", "synth.npz"))
X = torch.from_numpy(gd["mgcep512_X"]).to(DEV)
m32 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, c=3, n_iter=5, device=DEV)
y32 = m32(X.float()).cpu().numpy().astype(np.float64)
print("mgcep speech f32: max |err|:", float(np.abs(y32 - gd["mgcep512_c3_5"]).max()), " max |ref|:", float(np.abs(gd["mgcep512_c3_5"]).max()))
m0 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=0, n_iter=10, device=DEV)
y0 = m0(X.float()).cpu().numpy().astype(np.float64)
print("mgcep gamma=0 route f32: max |err|:", float(np.abs(y0 - gd["mcep512"]).max()), " max rel:", float((np.abs(y0 - gd["mcep512"]) / (np.abs(gd["mcep512"]) + 1e-30)).max()))
# mgcep random spectra (tests/test_gpu_configs.py::test_mgcep_fused_spectrum_arithmetic...)
xr = torch.randn(8, 4000, generator=torch.Generator().manual_seed(11), dtype=torch.float64)
Xr = dsp.STFT(400, 80, 512, device=DEV)(xr.to(DEV, torch.float32))
for gamma, M in ((-0.5, 24), (-0.25, 30), (-1 / 3, 12)):
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=M, alpha=0.42, gamma=gamma, n_iter=4, device=DEV)
    a = mg(Xr)
    refm = O.mgcep(Xr.double().cpu().numpy(), M, 0.42, gamma, 4)
    print(f"mgcep f32 gamma={gamma:.3f} M={M}: max |err| / max |ref|:", float(np.abs(a.double().cpu().numpy() - refm).max() / np.abs(refm).max()))
# smoke-sized mcep
xs2 = torch.randn(8, 4000, generator=torch.Generator().manual_seed(0))
X2 = stft(xs2.to(DEV))
Xref = O.stft(xs2.double().numpy(), 400, 80, 512)
print("smoke mcep: max |err|:", float(np.abs(mcep(X2).detach().cpu().numpy() - O.mcep(Xref, 24, 0.42, 10)).max()))
# LPC branch, config 4 (tests/test_gpu_parity.py::test_lpc_config4_batch1024_sampled, tests/test_gpu_lpc_fused.py): the float32 kernels
# against the float64 oracle on the same float32 samples -- lag sums as 3-term binary16 splits on the matrix pipe (default) and as
# exact float64 sums (DSA_LPC_EXACT_LAGSUMS); and the one-launch gradient against the float64 module chain
from diffsptk_amd import ops  # noqa: E402
xl = torch.randn(1024, 16000, generator=torch.Generator().manual_seed(0))
wl = dsp.Window(400, device=DEV).window
sel = slice(0, 1024, 171)
refl = O.frame_window_lpc(xl[sel].double().numpy())
for name, exact in (("matrix-pipe lag sums (default)", False), ("exact float64 lag sums", True)):
    a = ops.frame_window_lpc(xl.to(DEV), wl, 400, 80, 24, 1e-5, exact_lag_sums=exact)[sel].double().cpu().numpy()
    print(f"LPC config 4, {name}: max |err| gain {np.abs(a[..., 0] - refl[..., 0]).max():.3e}  coefficients {np.abs(a[..., 1:] - refl[..., 1:]).max():.3e}"
          f"  max rel (coefficients above 1e-2) {(np.abs(a - refl) / np.abs(refl))[np.abs(refl) > 1e-2].max():.3e}")
frm, wn = dsp.Frame(400, 80), dsp.Window(400, device=DEV)
fl = dsp.fuse(frm, wn, dsp.LPC(400, 24, eps=1e-5, device=DEV))
xg = xl[:8].to(DEV).requires_grad_(True)
gy = torch.randn(8, 200, 25, generator=torch.Generator().manual_seed(1)).to(DEV)
(fl(xg) * gy).sum().backward()
xg64 = xl[:8].double().to(DEV).requires_grad_(True)
ch = dsp.LPC(400, 24, eps=1e-5, device=DEV, dtype=torch.float64)(dsp.Window(400, device=DEV, dtype=torch.float64)(frm(xg64)))
(ch * gy.double()).sum().backward()
print(f"LPC one-launch gradient vs float64 chain: max |err| / max |ref| {float((xg.grad.double() - xg64.grad).abs().max() / xg64.grad.abs().max()):.3e}")
