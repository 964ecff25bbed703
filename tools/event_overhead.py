import os, sys, torch, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
dev="cuda"
x=torch.randn(1024,16000,device=dev)
stft=dsp.STFT(400,80,512,device=dev); mcep=dsp.MelCepstralAnalysis(fft_length=512,cep_order=24,alpha=0.42,n_iter=10,device=dev)
with torch.no_grad():
    X=stft(x)
    for _ in range(20): mcep(stft(x))
    torch.cuda.synchronize()
    res={"null":[], "stft":[], "stft2":[]}
    for rep in range(30):
        for kind in ("null","stft","stft2"):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            mcep(X)
            e0.record()
            if kind=="stft": stft(x)
            if kind=="stft2": stft(x); stft(x)
            e1.record()
            mcep(X)
            torch.cuda.synchronize()
            res[kind].append(e0.elapsed_time(e1)*1e3)
for k,v in res.items(): print(k, "median %.1f us  min %.1f  max %.1f"%(statistics.median(v),min(v),max(v)))
