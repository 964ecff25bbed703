"""The glogx pass alone (dsa_mcep_newton_glogx_h, 2048 / 49, 10 steps) and the sweep's launch with / without the in-place accumulation: one line."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import ops
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
out = []
for K, n, F in ((1025, 50, 102400), (513, 35, 122880)):
    g = torch.Generator().manual_seed(0)
    m = dsp.MelCepstralAnalysis(fft_length=2 * (K - 1), cep_order=n - 1, alpha=0.55, n_iter=1, device=dev)
    images = ops.mcep_resid_bwd_images(m.D, m.E)
    logx = (torch.randn(F, K, generator=g) * 0.7).to(dev)
    mcs = (torch.randn(10, F, n, generator=g) * 0.02).to(dev)
    grts = torch.randn(10, F, 2 * n - 1, generator=g).to(dev)
    glogx = torch.zeros(F, K, device=dev); gmc = torch.empty(F, n, device=dev)
    t_pass = timeit(lambda: ops._call("dsa_mcep_newton_glogx_h", ops._p(logx), F, K, ops._p(mcs), n, ops._p(grts), 10, ops._p(images), ops._dtype_code(logx), ops._p(glogx), ops._stream()))
    t_acc = timeit(lambda: ops._call("dsa_mcep_newton_resid_h_bwd", ops._p(logx), F, K, ops._p(mcs[0]), n, ops._p(grts[0]), ops._p(images), ops._dtype_code(logx), ops._p(glogx), ops._p(gmc), ops._stream()))
    t_null = timeit(lambda: ops._call("dsa_mcep_newton_resid_h_bwd", ops._p(logx), F, K, ops._p(mcs[0]), n, ops._p(grts[0]), ops._p(images), ops._dtype_code(logx), None, ops._p(gmc), ops._stream()))
    out.append(f"K={K} n={n} F={F}: glogx pass {t_pass:.0f} us | sweep launch accumulating {t_acc:.0f} us, glogx=NULL {t_null:.0f} us")
print(" || ".join(out))
