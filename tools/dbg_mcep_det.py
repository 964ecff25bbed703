import os, sys, torch
sys.path.insert(0, "/root/repo")
import diffsptk_amd as dsp
x = torch.randn(1024, 16000, generator=torch.Generator().manual_seed(1)).to("cuda")
X = dsp.STFT(400, 80, 512).to("cuda")(x)
m = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10).to("cuda")
ref = m(X).clone()
nd = 0
for i in range(20):
    nd += int((m(X) != ref).sum().item())
print("mcep forward: non-identical elements over 20 reruns:", nd)
Xg = X[:256].clone().requires_grad_(True)
m(Xg).square().sum().backward()
g0 = Xg.grad.clone()
nd = 0
for i in range(10):
    Xg.grad = None
    m(Xg).square().sum().backward()
    nd += int((Xg.grad != g0).sum().item())
print("mcep backward: non-identical elements over 10 reruns:", nd)
