"""48 kHz analysis forward + backward: glogx in one pass after the sweep (dsa_mcep_newton_glogx_h) against the in-place accumulation."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
g = torch.Generator().manual_seed(0)
for nfft, M, F in ((2048, 49, 102400), (2048, 49, 12800), (1024, 34, 122880)):
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
    X = (torch.randn(F, nfft // 2 + 1, generator=g).square() + 0.05).to(dev).requires_grad_(True)
    def fb():
        X.grad = None
        m(X).sum().backward()
    for rep in range(2):
        for mode in ("1", "0"):
            os.environ["DSA_MCEP_GLOGX_PASS"] = mode
            print(f"{nfft}/{M} F={F} glogx pass={mode}: forward + backward {timeit(fb):.2f} ms", flush=True)
