import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
g = torch.Generator().manual_seed(0)
for nfft, M in ((2048, 49), (1024, 34)):
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
    for F in (4096, 8192, 16384, 24576, 32768, 49152, 65536):
        X = (torch.randn(F, nfft // 2 + 1, generator=g).square() + 0.05).to(dev).requires_grad_(True)
        def fb():
            X.grad = None
            m(X).sum().backward()
        r = {}
        for rep in range(2):
            for mode in ("1", "0"):
                os.environ["DSA_MCEP_GLOGX_PASS"] = mode
                r.setdefault(mode, []).append(timeit(fb))
        print(f"{nfft}/{M} F={F}: pass {min(r['1']):.3f} ms, accumulating {min(r['0']):.3f} ms", flush=True)
