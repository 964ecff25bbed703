"""One line of timings (us per call) of every kernel family the round-6 purge of crossed packed float32 instructions touched, for
A/B runs against the round-5 library (tools/ab_libs.sh <out> <rounds> tools/time_purge_ab.py build/lib_r5.so product)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
x = torch.randn(1024, 16000, generator=g).to(dev)
st = dsp.STFT(400, 80, 512, device=dev)
fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, use_power=True, device=dev)
fused_fb = dsp.fuse(st, fb)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
out = {}
with torch.no_grad():
    S = st(x)
    out["stft_fwd"] = timeit(lambda: st(x), 50)
    out["stft_fbank"] = timeit(lambda: fused_fb(x), 50)
    out["fbank_mod"] = timeit(lambda: fb(S))
    out["fused_mcep"] = timeit(dsp.fuse(st, mcep).__call__ if False else (lambda: dsp.fuse(st, mcep)(x)))
xg = x.clone().requires_grad_(True); y = st(xg); gy = torch.randn_like(y)
out["stft_bwd"] = timeit(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True))
Sg = S[:256].clone().requires_grad_(True); yb = fb(Sg); gb = torch.randn_like(yb)
out["fbank_bwd"] = timeit(lambda: torch.autograd.grad(yb, Sg, gb, retain_graph=True))
del y, gy, xg
with torch.no_grad():
    mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=10, device=dev)
    Sq = S[:256].contiguous()
    out["mgcep24"] = timeit(lambda: mg(Sq), 10)
    mg17 = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=17, alpha=0.42, gamma=-0.5, n_iter=5, device=dev)
    out["mgcep17"] = timeit(lambda: mg17(Sq), 5)
    x48 = torch.randn(64, 48000, generator=g).to(dev)
    for fl, fp, nfft, M, a in ((1200, 240, 2048, 49, 0.55), (800, 200, 1024, 34, 0.55)):
        X48 = dsp.STFT(fl, fp, nfft, device=dev)(x48)
        m48 = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=M, alpha=a, n_iter=10, device=dev)
        out[f"mcep{nfft}"] = timeit(lambda: m48(X48), 10)
    mc = mcep(S[:256])[:, :200]
    xs = x[:256].contiguous()
    for mode in ("multi-stage", "single-stage"):
        ml = dsp.MLSA(24, 80, alpha=0.42, mode=mode, device=dev)
        out["mlsa_" + mode[:5]] = timeit(lambda: ml(xs, mc), 5)
print((sys.argv[1] if len(sys.argv) > 1 else "") + " | ".join(f"{k} {v:.1f}" for k, v in out.items()))
