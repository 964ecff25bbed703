import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
dev = "cuda"
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
fused = dsp.fuse(stft, mcep)
for B in (8, 64, 163, 164, 200, 1024):
    x = torch.randn(B, 16000, device=dev, generator=torch.Generator(device=dev).manual_seed(B))
    with torch.no_grad():
        two = mcep(stft(x)).reshape(-1, 25)
        for rep in range(2):
            one = fused(x).reshape(-1, 25)
            bad = ((one - two).abs().amax(-1) > 1e-6).nonzero().flatten().cpu()
            print(f"B={B} rep={rep} tiles={B*200//16} bad frames={bad.numel()}", flush=True)
            if bad.numel():
                t = bad // 16
                print("  tiles:", sorted(set(t.tolist()))[:40])
                print("  frame-in-tile:", sorted(set((bad % 16).tolist())))
                print("  n in utterance:", sorted(set((bad % 200).tolist()))[:60])
                print("  first:", bad[:20].tolist(), float((one - two).abs().max()))
