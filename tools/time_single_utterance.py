import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = torch.device("cuda", 0)
pcm = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/datawav.npz"))["pcm"]
x = torch.from_numpy(pcm.astype(np.float32) / 32768.0).to(dev)
stft = dsp.STFT(400, 80, 512, device=dev); mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
fused = dsp.fuse(stft, mcep)
def t(fn, n=300):
    with torch.no_grad():
        for _ in range(30): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); thr = (time.perf_counter() - t0) / n
        lat = []
        for _ in range(50):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    return thr * 1e6, float(np.median(lat)) * 1e6
for name, fn in (("stft(x)", lambda: stft(x)), ("mcep(stft(x))", lambda: mcep(stft(x))), ("fuse(stft, mcep)(x)", lambda: fused(x))):
    thr, lat = t(fn)
    print(f"data.wav (19 200 samples, 240 frames) {name}: {thr:.1f} us per call back to back, {lat:.1f} us call-to-result (median)")
g = dsp.Graphed(lambda z: fused(z), x)
thr, lat = t(lambda: g(x))
print(f"  the same as a replayed HIP graph (dsp.Graphed): {thr:.1f} us per call back to back, {lat:.1f} us call-to-result")
