import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import _lib, ops
dev = "cuda"
stft = dsp.STFT(400, 80, 512, device=dev)
B = 1024
x = torch.randn(B, 16000, device=dev, generator=torch.Generator(device=dev).manual_seed(B))
T, N, F = 16000, 200, B * 200
scratch = torch.zeros(_lib.SCRATCH_BYTES, dtype=torch.uint8, device=dev)
with torch.no_grad():
    X2 = stft(x).view(F, 257)
    for n_iter in (0, 1, 3, 10):
        mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=n_iter, device=dev)
        images = ops.mcep_images(mcep.G, mcep.D, mcep.E, 512, 24)
        for rep in range(2):
            mc = torch.empty(F, 25, device=dev)
            X = torch.full((F, 257), float("nan"), device=dev)
            ops._call("dsa_stft_mcep_fwd", x.data_ptr(), B, T, 400, 80, 512, stft.window.data_ptr(), stft.twiddle.data_ptr(), 1, 1e-9, 24, n_iter,
                      mcep.G.data_ptr(), mcep.D.data_ptr(), mcep.E.data_ptr(), mcep.alpha_vector.data_ptr(), _lib.F32, _lib.ALGO_AUTO,
                      images.data_ptr(), scratch.data_ptr(), mc.data_ptr(), None, X.data_ptr(), ops._stream())
            torch.cuda.synchronize()
            bad = (X != X2).any(-1).nonzero().flatten().cpu()
            print(f"n_iter={n_iter} rep={rep}: bad X frames {bad.numel()}  first tiles {sorted(set((bad // 16).tolist()))[:8]}", flush=True)
