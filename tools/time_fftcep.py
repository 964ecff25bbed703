"""Timing of CepstralAnalysis with refinement iterations (n_iter = 3) at 51 200 frames, forward and forward + backward."""
import sys,os,torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev="cuda"
X=(torch.randn(51200,257,generator=torch.Generator().manual_seed(0)).square()+0.1).to(dev)
def gpu_time(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ca=dsp.CepstralAnalysis(fft_length=512,cep_order=24,n_iter=3,device=dev)
with torch.no_grad():
    print("fftcep n_iter=3 ms", gpu_time(lambda: ca(X)))
def fb():
    xg=X.clone().requires_grad_(True); ca(xg).sum().backward()
print("fwd+bwd ms", gpu_time(fb,3))
