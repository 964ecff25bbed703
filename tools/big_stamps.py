"""Cycle stamps of one Newton step (step 2 of workgroup 0's first tile) of the 48 kHz one-launch kernel, narrow and wide tile shapes
(a library built with -DDSA_BIG_STAMPS: tools/build_variant.sh build/lib_bigstamps.so mcep_mfma.hip -DDSA_BIG_STAMPS)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
g = torch.Generator().manual_seed(0)
names = ["top barrier + B operands + first staging", "stage loop", "partial sums", "records", "solve (call)", "records 2", "solve 2"]
for nfft, M in ((2048, 49), (1024, 34)):
    for F in (12800, 102400):
        X = (torch.randn(F, nfft // 2 + 1, generator=g).square() + 0.05).cuda()
        m = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=0.55, n_iter=10, device="cuda")
        for wide in ("0", "1"):
            os.environ["DSA_MCEP_BIG_WIDE"] = wide
            with torch.no_grad():
                y = m(X); y = m(X)
            for w in ((0, 8) if wide == "0" else (0,)):
                print(f"{nfft}/{M} F={F} {'wide' if wide == '1' else 'narrow'} wave {'0' if w == 0 else '4 (odd stages)'}:",
                      "  ".join(f"{names[i - 1]} {int(y[w, i])}" for i in range(1, 8)), " total", int(sum(float(y[w, i]) for i in range(1, 8))))
                pn = ["issue fetches", "first chain", "t, max, exp, split", "second chain + sums", "(second stage)", "stage the next pair", "barrier"]
                print("      pair 4:", "  ".join(f"{pn[i - 9]} {int(y[w, i])}" for i in range(9, 16)))
