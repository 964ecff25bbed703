"""The TWIN-workgroup shape of dsa_mcep_newton_steps (csrc/mcep_big4_f16.h: two four-wave workgroups per CU half a step apart) against the
eight-wave wide tiles, the narrow tiles and the two launches per step: bit for bit at every batch size / order / iteration count, and the
call times over the stagger.   usage: python tools/check_big_twin.py [quick]"""
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
def run(m, X, big, wide, twin):
    os.environ["DSA_MCEP_BIG"] = big
    if wide is None: os.environ.pop("DSA_MCEP_BIG_WIDE", None)
    else: os.environ["DSA_MCEP_BIG_WIDE"] = wide
    os.environ["DSA_MCEP_BIG_TWIN"] = twin
    with torch.no_grad():
        return m(X)
g = torch.Generator().manual_seed(0)
bad = 0
for M, K in ((49, 1025), (34, 513), (32, 1025), (46, 1025), (54, 1025), (40, 513), (37, 257)):
    for F in (1, 15, 64, 65, 127, 129, 3217, 20011, 36000) + (() if quick else (70001,)):
        X = (torch.randn(F, K, generator=g).square() + 0.05).to(dev)
        for n_iter in (1, 2, 10):
            m = dsp.MelCepstralAnalysis(fft_length=2 * (K - 1), cep_order=M, alpha=0.55, n_iter=n_iter, device=dev)
            a = run(m, X, "0", "0", "0"); ka = _lib.last_kernel()
            w = run(m, X, "2", "1", "0")
            t = run(m, X, "2", "1", "1"); kt = _lib.last_kernel()
            t2 = run(m, X, "2", "1", "1")
            os.environ["DSA_MCEP_BIG_STAGGER"] = "0"
            t3 = run(m, X, "2", "1", "1")
            os.environ.pop("DSA_MCEP_BIG_STAGGER", None)
            d = run(m, X, "2", None, "1")                   # the plan
            eq = torch.equal(t, t2) and torch.equal(t, t3) and torch.equal(a, t) and torch.equal(w, t) and torch.equal(d, t) and kt == "mcep_big_newton"
            if not eq:
                bad += 1
                print(f"M={M} K={K} F={F} n_iter={n_iter}: {ka} / {kt}: max |two-launch - twin| {float((a - t).abs().max()):.3e} |wide - twin| "
                      f"{float((w - t).abs().max()):.3e} |plan - twin| {float((d - t).abs().max()):.3e} (finite {bool(torch.isfinite(t).all())}, repeat equal {torch.equal(t, t2)})")
print("mismatching cases:", bad)
for fl, fp, nfft, M in ((1200, 240, 2048, 49), (800, 200, 1024, 34)):
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
    for B in ((100, 164, 200, 330, 512, 1024) if not quick else (164, 512)):
        x = torch.randn(B, 48000, generator=g).to(dev)
        with torch.no_grad():
            X = dsp.STFT(fl, fp, nfft, device=dev)(x)
        for rep in range(2):
            rows = [("0", "0", "0", None, "two launches per step"), ("2", "0", "0", None, "narrow"), ("2", "1", "0", None, "wide"), ("2", None, "0", None, "planned, wide")]
            rows += [("2", "1", "1", s, f"twin, stagger {s}") for s in ("0", "2", "4", "6", "8", "10", "14")]
            rows += [("2", None, "1", None, "planned, twin")]
            for big, wide, twin, stag, name in rows:
                if stag is None: os.environ.pop("DSA_MCEP_BIG_STAGGER", None)
                else: os.environ["DSA_MCEP_BIG_STAGGER"] = stag
                run(m, X, big, wide, twin)
                with torch.no_grad():
                    t = timeit(lambda: m(X), 5)
                print(f"{nfft} / {M}: B={B} ({X.shape[0] * X.shape[1]} frames) {name}: {t:.1f} us per analysis ({_lib.last_kernel()})")
for k in ("DSA_MCEP_BIG_WIDE", "DSA_MCEP_BIG_STAGGER", "DSA_MCEP_BIG_TWIN", "DSA_MCEP_BIG"): os.environ.pop(k, None)
