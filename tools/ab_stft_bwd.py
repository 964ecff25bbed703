"""A/B of the packed STFT backward (csrc/stft_bwd_pk.h) against the two-kernel path: run twice, DSA_STFT_BWD_PK=1 / 0,
each run saves its gradients; the second run compares.  python tools/ab_stft_bwd.py <tag> [<other tag>]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import _lib
dev = "cuda"
tag = sys.argv[1]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
out = {}
g = torch.Generator().manual_seed(1)
for name, shape, center in (("a", (3, 16000), True), ("b", (2, 5, 4001), True), ("c", (7, 803), False), ("d", (1, 401), True),
                            ("e", (64, 16000), True), ("f", (5, 79), True), ("g", (2, 48000), False)):
    x = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
    st = dsp.STFT(400, 80, 512, center=center, device=dev)
    y = st(x)
    wt = torch.randn(y.shape, generator=g).to(dev)
    (gx,) = torch.autograd.grad(y, x, wt)
    out["pow_" + name] = gx.cpu(); out["pow_k_" + name] = _lib.last_kernel()
    stc = dsp.STFT(400, 80, 512, center=center, out_format="complex", device=dev)
    yc = stc(x)
    wc = torch.randn(yc.shape + (2,), generator=g).to(dev)
    (gc,) = torch.autograd.grad(yc, x, torch.view_as_complex(wc))
    out["cpx_" + name] = gc.cpu(); out["cpx_k_" + name] = _lib.last_kernel()
    if center:
        ist = dsp.ISTFT(400, 80, 512, device=dev)
        with torch.no_grad():
            out["ist_" + name] = ist(yc.detach(), out_length=shape[-1]).cpu(); out["ist_k_" + name] = _lib.last_kernel()
torch.save(out, f"/tmp/ab_stft_bwd_{tag}.pt")
x = torch.randn(1024, 16000, device=dev)
st = dsp.STFT(400, 80, 512, device=dev)
xg = x.clone().requires_grad_(True)
y = st(xg); gg = torch.ones_like(y)
t = timeit(lambda: torch.autograd.grad(y, xg, gg, retain_graph=True))
x2 = x[:256].clone().requires_grad_(True); y2 = st(x2); g2 = torch.ones_like(y2)
t2 = timeit(lambda: torch.autograd.grad(y2, x2, g2, retain_graph=True))
stc = dsp.STFT(400, 80, 512, out_format="complex", device=dev); ist = dsp.ISTFT(400, 80, 512, device=dev)
with torch.no_grad():
    Z = stc(x); ti = timeit(lambda: ist(Z))
print(f"[{tag}] kernels {out['pow_k_a']} / {out['cpx_k_a']} / {out['ist_k_a']}: stft bwd B=1024 {t*1e3:.1f} us, B=256 {t2*1e3:.1f} us | istft B=1024 {ti*1e3:.1f} us")
if len(sys.argv) > 2:
    ref = torch.load(f"/tmp/ab_stft_bwd_{sys.argv[2]}.pt")
    for k in out:
        if "_k_" in k: continue
        a, b = out[k], ref[k]
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
        print(f"  {k}: max |diff| / max |ref| = {err:.2e}  finite={bool(torch.isfinite(a).all())}")
