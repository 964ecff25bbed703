"""The wide tile shape of dsa_mcep_newton_steps (128 frames per workgroup, a wave per 16-frame group) against the narrow one (64 frames, two
waves per group) and the two launches per step: bit for bit at every batch size / order / iteration count, and the call times.
usage: python tools/check_big_wide.py [quick]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
quick = len(sys.argv) > 1
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
def run(m, X, big, wide):
    os.environ["DSA_MCEP_BIG"] = big
    os.environ["DSA_MCEP_BIG_WIDE"] = wide
    with torch.no_grad():
        return m(X)
g = torch.Generator().manual_seed(0)
bad = 0
for M, K in ((49, 1025), (34, 513), (32, 1025), (46, 1025), (54, 1025), (40, 513)):
    for F in (1, 15, 64, 65, 127, 129, 3217, 12800, 20011, 36000):
        X = (torch.randn(F, K, generator=g).square() + 0.05).to(dev)
        for n_iter in (1, 2, 10):
            m = dsp.MelCepstralAnalysis(fft_length=2 * (K - 1), cep_order=M, alpha=0.55, n_iter=n_iter, device=dev)
            a = run(m, X, "0", "0"); ka = _lib.last_kernel()
            b = run(m, X, "2", "0"); kb = _lib.last_kernel()
            c = run(m, X, "2", "1"); kc = _lib.last_kernel()
            c2 = run(m, X, "2", "1")
            os.environ.pop("DSA_MCEP_BIG_WIDE", None)
            with torch.no_grad():
                d = m(X)                                    # the plan
            eq = torch.equal(c, c2) and torch.equal(a, c) and torch.equal(b, c) and torch.equal(d, c) and kc == "mcep_big_newton"
            if not eq:
                bad += 1
                print(f"M={M} K={K} F={F} n_iter={n_iter}: {ka} / {kb} / {kc}: max |two-launch - wide| {float((a - c).abs().max()):.3e} |narrow - wide| "
                      f"{float((b - c).abs().max()):.3e} (finite {bool(torch.isfinite(c).all())}, repeat equal {torch.equal(c, c2)})")
print("mismatching cases:", bad)
for fl, fp, nfft, M in ((1200, 240, 2048, 49), (800, 200, 1024, 34)):
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
    for B in ((64, 100, 200, 330, 512, 1024) if not quick else (100, 512)):
        x = torch.randn(B, 48000, generator=g).to(dev)
        with torch.no_grad():
            X = dsp.STFT(fl, fp, nfft, device=dev)(x)
        for rep in range(2):
            for big, wide, name in (("0", "0", "two launches per step"), ("2", "0", "one launch, narrow"), ("2", "1", "one launch, wide"), ("2", "", "planned")):
                os.environ["DSA_MCEP_BIG"] = big
                if wide: os.environ["DSA_MCEP_BIG_WIDE"] = wide
                else: os.environ.pop("DSA_MCEP_BIG_WIDE", None)
                with torch.no_grad():
                    t = timeit(lambda: m(X), 5)
                print(f"{nfft} / {M}: B={B} ({X.shape[0] * X.shape[1]} frames) {name}: {t:.1f} us per analysis ({_lib.last_kernel()})")
os.environ.pop("DSA_MCEP_BIG_WIDE", None)
os.environ["DSA_MCEP_BIG"] = "2"
