"""Mel-generalized cepstral analysis (gamma = -0.5, 10 iterations, 51 200 frames) wall time per call, for A/B runs."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
x = torch.randn(256, 16000, device="cuda")
st = dsp.STFT(400, 80, 512, device="cuda")
mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, gamma=-0.5, n_iter=10, device="cuda")
with torch.no_grad():
    X = st(x)
    for _ in range(5): mg(X)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): mg(X)
    e1.record(); torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: mgcep {e0.elapsed_time(e1) / 10:.2f} ms")
