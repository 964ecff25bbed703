import sys,torch,os
sys.path.insert(0,"/root/repo" if os.path.isdir("/root/repo/diffsptk_amd") else ".")
import diffsptk_amd as dsp
dev="cuda"
def gpu_time(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
X=(torch.randn(51200,257,generator=torch.Generator().manual_seed(0)).square()+0.1).to(dev)
mc=(0.1*torch.randn(51200,25,generator=torch.Generator().manual_seed(1))).to(dev)
with torch.no_grad():
    mg=dsp.MelGeneralizedCepstralAnalysis(fft_length=512,cep_order=24,alpha=0.42,gamma=-0.5,n_iter=10,device=dev)
    print("mgcep ms", gpu_time(lambda: mg(X), 5))
    m=dsp.MelGeneralizedCepstrumToMelGeneralizedCepstrum(24,30,in_gamma=0,out_gamma=-0.5,device=dev)
    print("mgc2mgc 24->30 gamma 0->-0.5 ms", gpu_time(lambda: m(mc)))
    ml=dsp.MLSA(24,80,alpha=0.42,mode="multi-stage",device=dev)
    x=torch.randn(256,16000,device=dev)
    print("MLSA multi-stage ms", gpu_time(lambda: ml(x, mc.view(256,200,25)), 3))
