mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stft_bwd.py -q -x -k "lpc or acorr or levdur" 2>&1 | tail -12 > gpurun_out/r4f/pytest.txt
python - > gpurun_out/r4f/time.txt 2>&1 <<'P'
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import diffsptk_amd as dsp
from diffsptk_amd import ops, _lib
from oracle import oracle as O
dev="cuda"
x = torch.randn(1024, 16000, generator=torch.Generator().manual_seed(0))
xd = x.to(dev)
w = dsp.Window(400, device=dev).window
def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
a = ops.frame_window_lpc(xd, w, 400, 80, 24, 1e-5); k=_lib.last_kernel()
print("kernel", k, "ms", t(lambda: ops.frame_window_lpc(xd, w, 400, 80, 24, 1e-5)))
sel = slice(0, 1024, 171)
ref = O.frame_window_lpc(x[sel].double().numpy())
err = np.abs(a[sel].cpu().numpy() - ref)
print("max |a - oracle| randn", err.max(), " worst vs (1e-4 + 1e-4|ref|):", (err / (1e-4 + 1e-4*np.abs(ref))).max())
print("reproducible:", bool(torch.equal(a, ops.frame_window_lpc(xd, w, 400, 80, 24, 1e-5))))
P
cat gpurun_out/r4f/pytest.txt gpurun_out/r4f/time.txt
