"""Which stage of the mel-generalized analysis at an order other than 24 rounds a frame differently when fewer frames travel with it:
every sub-operator of modules/mgcep.py on F = 200 frames against the same frames alone / in small groups, bit for bit."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import ops, _lib
from diffsptk_amd.modules.mgcep import _Gnorm, _Ignorm
dev = "cuda"
for M in (17, 30):
    gen = torch.Generator().manual_seed(M)
    X = (torch.randn(200, 257, generator=gen).square() + 0.1).to(dev)
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=M, alpha=0.42, gamma=-0.5, n_iter=2, device=dev)
    b1 = (0.05 * torch.randn(200, M, generator=gen)).to(dev)
    b = (0.05 * torch.randn(200, M + 1, generator=gen)).to(dev); b[:, 0] = b[:, 0].abs() + 1
    stages = {
        "mm Pr": lambda x, b1, b: ops.MatmulRowsFn.apply(x, mg.Pr),
        "mm R1": lambda x, b1, b: ops.MatmulRowsFn.apply(x, mg.R1),
        "ignorm": lambda x, b1, b: _Ignorm._forward(b, gamma=-1),
        "b2mc": lambda x, b1, b: mg.b2mc(b),
        "gc2gc": lambda x, b1, b: mg.gc2gc(b),
        "mc2b": lambda x, b1, b: mg.mc2b(b),
        "gnorm": lambda x, b1, b: _Gnorm._forward(b, gamma=-0.5),
        "mgcep_spectra": lambda x, b1, b: ops.mgcep_spectra(x, b1, mg.Cr, mg.Ci, -0.5).transpose(0, 1),
    }
    if mg.step_images is not None:
        stages["mgcep_step"] = lambda x, b1, b: torch.cat(ops.mgcep_step(x, b1, mg.step_images, -0.5), -1)
    def solve(x, b1, b):
        pt = ops.MatmulRowsFn.apply(x, mg.Pr); r = ops.MatmulRowsFn.apply(x, mg.R1)
        qt = torch.zeros(x.size(0), 2 * M - 1, device=dev)
        return ops.ThSolveFn.apply(pt, qt, r[..., 1:].contiguous())
    stages["thsolve"] = solve
    stages["whole (no grad)"] = lambda x, b1, b: mg(x)
    def whole_grad(x, b1, b):
        xs = x.clone().requires_grad_(True)
        with torch.enable_grad():
            y = mg(xs)
        return y.detach()
    stages["whole (grad mode)"] = whole_grad
    with torch.no_grad():
        for name, fn in stages.items():
            ref = fn(X, b1, b)
            k = _lib.last_kernel()
            bad = []
            for F in (1, 16, 63, 64, 65):
                out = fn(X[:F].contiguous(), b1[:F].contiguous(), b[:F].contiguous())
                if not torch.equal(out, ref[:F]):
                    bad.append((F, float((out - ref[:F]).abs().max())))
            print(f"M={M} {name:20s} kernel={k}: {'invariant' if not bad else bad}")
