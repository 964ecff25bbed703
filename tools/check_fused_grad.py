import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd.utils import tables
from oracle import torch_port as TP
dev="cuda"
g=torch.Generator().manual_seed(3)
for C,sr,use_power,gamma,mf in [(40,16000,True,0.0,False),(40,16000,False,0.0,False),(80,22050,True,-0.3,False),(24,16000,False,0.4,False),(40,16000,False,0.0,True)]:
    x=torch.randn(3,8000,generator=g)*torch.tensor([1e-2,1.0,30.0]).view(3,1)
    stft=dsp.STFT(400,80,512,device=dev)
    if mf:
        an=dsp.MFCC(fft_length=512,mfcc_order=12,n_channel=C,sample_rate=sr,lifter=22,device=dev)
    else:
        an=dsp.MelFilterBankAnalysis(fft_length=512,n_channel=C,sample_rate=sr,use_power=use_power,gamma=gamma,device=dev)
    fused=dsp.fuse(stft,an)
    xa=x.to(dev).requires_grad_(True); ya=fused(xa); cot=torch.randn(ya.shape,generator=g).to(dev)
    (ga,)=torch.autograd.grad(ya,xa,cot); pa=fused.last_path
    xb=x.to(dev).requires_grad_(True); yb=an(stft(xb)); (gb,)=torch.autograd.grad(yb,xb,cot)
    e2=((ga-gb).abs().amax(-1)/gb.abs().amax(-1)).max().item()
    msg=f"C={C} pow={use_power} gamma={gamma} mfcc={mf}: path {pa}; vs two-stage {e2:.2e}"
    if not mf:
        H=torch.from_numpy(np.asarray(tables.fbank_matrix(512,C,sr,0.0,None,"htk",None))).double()
        xr=x.double().requires_grad_(True)
        P=TP.stft_power(xr,400,80,512)
        s=(P if use_power else torch.sqrt(P))@H
        s=torch.clip(s,min=1e-5)
        y=torch.log(s) if gamma==0 else (torch.pow(s,gamma)-1)/gamma
        (y*cot.cpu().double()).sum().backward()
        e3=((ga.cpu().double()-xr.grad).abs().amax(-1)/xr.grad.abs().amax(-1)).max().item()
        e4=((gb.cpu().double()-xr.grad).abs().amax(-1)/xr.grad.abs().amax(-1)).max().item()
        msg+=f"; fused vs f64 {e3:.2e}; two-stage vs f64 {e4:.2e}"
    print(msg)
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
x=torch.randn(1024,16000,device=dev)
stft=dsp.STFT(400,80,512,device=dev)
for up in (True,False):
    fb=dsp.MelFilterBankAnalysis(fft_length=512,n_channel=40,sample_rate=16000,use_power=up,device=dev)
    fused=dsp.fuse(stft,fb)
    def fa():
        xg=x.detach().requires_grad_(True); y=fused(xg); y.backward(torch.ones_like(y))
    def fb2():
        xg=x.detach().requires_grad_(True); y=fb(stft(xg)); y.backward(torch.ones_like(y))
    print(f"use_power={up}: fwd+bwd fused {timeit(fa):.0f} us, two-stage {timeit(fb2):.0f} us")
