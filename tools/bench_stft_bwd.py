import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev="cuda"
x=torch.randn(1024,16000,device=dev)
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
st=dsp.STFT(400,80,512,device=dev)
xg=x.clone().requires_grad_(True)
y=st(xg); g=torch.ones_like(y)
t=timeit(lambda: torch.autograd.grad(y,xg,g,retain_graph=True))
stc=dsp.STFT(400,80,512,out_format="complex",device=dev); ist=dsp.ISTFT(400,80,512,device=dev)
with torch.no_grad():
    Z=stc(x); ti=timeit(lambda: ist(Z))
print(f"stft bwd {t:.3f} ms | istft {ti:.3f} ms")
