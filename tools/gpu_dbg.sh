#!/bin/bash
mkdir -p gpurun_out
python - > gpurun_out/dbg.log 2>&1 <<'PY'
import torch, traceback
import diffsptk_amd as dsp
from diffsptk_amd import ops
dev="cuda"
orig = ops.ThSolveFn.forward
x = torch.randn(8, 4000, generator=torch.Generator().manual_seed(11), dtype=torch.float64)
for dt in (torch.float64, torch.float32):
    X = dsp.STFT(400, 80, 512, dtype=dt, device=dev)(x.to(dev, dt))
    for gamma, M in ((-0.5, 24), (-0.25, 30), (-1 / 3, 12)):
        mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=512, cep_order=M, alpha=0.42, gamma=gamma, n_iter=4, dtype=dt, device=dev)
        for grad in (False, True):
            Xi = X.clone().requires_grad_(grad)
            try:
                out = mg(Xi)
                print(dt, gamma, M, grad, "ok", tuple(out.shape), float(out.detach().abs().max()))
            except Exception as e:
                print(dt, gamma, M, grad, "FAILED", e)
                import inspect
                tb = traceback.extract_tb(e.__traceback__)
                fr = e.__traceback__
                while fr.tb_next: fr = fr.tb_next
                loc = fr.tb_frame.f_locals
                print({k: tuple(v.shape) for k, v in loc.items() if hasattr(v, "shape")})
PY
cat gpurun_out/dbg.log
