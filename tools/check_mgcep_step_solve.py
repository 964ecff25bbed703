"""dsa_mgcep_step_solve (one launch, binary16-split chains) against dsa_mgcep_step + dsa_thsolve_update_fwd (float32 matrix instructions)
and against the float64 oracle: the step's outputs and the whole analysis (gamma = -0.5 and -1/3, speech-like and random spectra)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import ops
from oracle import oracle as O
DEV = "cuda"
x = torch.randn(16, 8000, generator=torch.Generator().manual_seed(5))
X = dsp.STFT(400, 80, 512, device=DEV)(x.to(DEV)).reshape(-1, 257)[:1595]
for gamma, n_iter in ((-0.5, 4), (-1 / 3, 10), (-0.25, 2)):
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=gamma, n_iter=n_iter, device=DEV)
    # one step from a realistic b1
    b1 = 0.05 * torch.randn(X.size(0), 24, generator=torch.Generator().manual_seed(1)).to(DEV)
    pt, qt, r = ops.mgcep_step(X, b1, mg.step_images, gamma)
    ref_b = ops.thsolve_update(pt, qt, r, b1)
    new_b, new_r = ops.mgcep_step_solve(X, b1, mg.step_images_h, gamma)
    print(f"gamma {gamma:.3f}: step r max rel {float((new_r - r).abs().max() / r.abs().max()):.3e}  b1 max |diff| {float((new_b - ref_b).abs().max()):.3e} (max |b1| {float(ref_b.abs().max()):.3f})")
    with torch.no_grad():
        y1 = mg(X)
        os.environ["DSA_MGCEP_STEP_SOLVE"] = "0"
        y0 = mg(X)
        del os.environ["DSA_MGCEP_STEP_SOLVE"]
    ref = O.mgcep(X.double().cpu().numpy(), 24, 0.42, gamma, n_iter)
    e1 = np.abs(y1.double().cpu().numpy() - ref).max() / np.abs(ref).max()
    e0 = np.abs(y0.double().cpu().numpy() - ref).max() / np.abs(ref).max()
    print(f"   analysis ({n_iter} steps): one-launch step vs float64 oracle {e1:.3e}   two-launch step {e0:.3e}   between them {float((y1 - y0).abs().max() / y0.abs().max()):.3e}")
