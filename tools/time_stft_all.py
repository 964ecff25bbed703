"""One line of STFT-family timings (us per launch, 204 800 frames unless noted) for A/B runs of library builds."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev = "cuda"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(1024, 16000, device=dev)
st = dsp.STFT(400, 80, 512, device=dev)
fb = dsp.MelFilterBankAnalysis(fft_length=512, n_channel=40, sample_rate=16000, use_power=True, device=dev)
fused = dsp.fuse(st, fb)
with torch.no_grad():
    t_f = timeit(lambda: st(x)); t_fb = timeit(lambda: fused(x)); t_f64 = timeit(lambda: st(x[:64]))
xg = x.clone().requires_grad_(True); y = st(xg); g = torch.randn_like(y)
t_b = timeit(lambda: torch.autograd.grad(y, xg, g, retain_graph=True))
x2 = x[:256].clone().requires_grad_(True); y2 = st(x2); g2 = torch.randn_like(y2)
t_b2 = timeit(lambda: torch.autograd.grad(y2, x2, g2, retain_graph=True))
stc = dsp.STFT(400, 80, 512, out_format="complex", device=dev); ist = dsp.ISTFT(400, 80, 512, device=dev)
with torch.no_grad():
    Z = stc(x); t_i = timeit(lambda: ist(Z))
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: fwd {t_f:.1f} | fwd B=64 {t_f64:.1f} | fused fbank {t_fb:.1f} | bwd {t_b:.1f} | bwd B=256 {t_b2:.1f} | istft {t_i:.1f}")
