"""Ablation timings of the twin-workgroup Newton kernel.  The ablations are COMPILE-TIME since the round's end (tools/build_variant.sh <out.so> mcep_mfma.hip
-DDSA_BIG_ABL=1|2|3: 1 no solve, 2 no products; results are garbage, times are not): run this script under tools/ab_libs.sh with those builds -- the
DSA_MCEP_BIG_ABL variable it sets is no longer read by the library (the numbers of profiles/r06_mcep_big_twin.txt were taken with the run-time switch)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
os.environ["DSA_MCEP_BIG"] = "2"; os.environ["DSA_MCEP_BIG_WIDE"] = "1"
for nfft, M in ((2048, 49), (1024, 34)):
    m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device=dev)
    for F in (16384, 32768, 65536):
        X = (torch.randn(F, nfft // 2 + 1, generator=g).square() + 0.05).to(dev)
        for twin in ("1", "0"):
            os.environ["DSA_MCEP_BIG_TWIN"] = twin
            for abl in ("0", "1", "2", "3"):
                if twin == "0" and abl != "0": continue
                os.environ["DSA_MCEP_BIG_ABL"] = abl
                for stag in (("0", "6") if twin == "1" else ("0",)):
                    os.environ["DSA_MCEP_BIG_STAGGER"] = stag
                    with torch.no_grad():
                        t = timeit(lambda: m(X))
                    print(f"{nfft}/{M} F={F} twin={twin} abl={abl} stagger={stag}: {t:.1f} us ({_lib.last_kernel()})")
