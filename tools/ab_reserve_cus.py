"""DSA_ALGO_RESERVE_CUS on a one-GPU box: the bench step (STFT -> mel-cepstrum, one launch, 1 024 utterances) with a stand-in for the
previous step's all-gather on a second stream -- tools/occupy.cpp: W workgroups that hold their CUs for D microseconds, ordered after
the step's kernel by an event; the compute stream waits for the stand-in of the step BEFORE the previous one (bench.py's deferred
gather).  Reports ms per step for n reserved CUs.

    hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/occupy.cpp -o build/liboccupy.so
    python tools/ab_reserve_cus.py [W = 16] [D = 350]
"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffsptk_amd as dsp  # noqa: E402
from diffsptk_amd import ops  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
D = float(sys.argv[2]) if len(sys.argv) > 2 else 350.0
dev = torch.device("cuda", 0)
occ = ctypes.CDLL(os.path.join(ROOT, "build", "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
occ.occupy_sleep.argtypes = [ctypes.c_int, ctypes.c_long, ctypes.c_void_p]
MODE = os.environ.get("STANDIN", "ordered")   # ordered: after the step's kernel, completed one step later (bench.py's deferred gather) |
                                               # free: launched on the second stream with no dependency either way | sleep: ordered, a kernel that only sleeps
sink = torch.zeros(4, dtype=torch.int32, device=dev)
x = torch.randn(1024, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
fused = dsp.fuse(stft, mcep)
side = torch.cuda.Stream()
ref = None


def run(n_reserve, with_standin, steps=200):
    global ref
    main = torch.cuda.current_stream()
    pending = []
    with torch.no_grad(), ops.reserve_cus(n_reserve):
        def step():
            y = fused(x)
            if with_standin and MODE == "free":
                occ.occupy(W, int(D * 1000), sink.data_ptr(), side.cuda_stream)
            elif with_standin:
                e = torch.cuda.Event()
                e.record(main)
                side.wait_event(e)
                if MODE == "sleep":
                    occ.occupy_sleep(W, int(D * 1000), side.cuda_stream)
                else:
                    occ.occupy(W, int(D * 1000), sink.data_ptr(), side.cuda_stream)
                g = torch.cuda.Event()
                g.record(side)
                pending.append(g)
                while len(pending) > (10 ** 9 if MODE == "nowait" else int(os.environ.get("DEPTH", "1"))):   # nowait: the analysis never waits for the stand-in
                    main.wait_event(pending.pop(0))
            return y
        for _ in range(30):
            y = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
    if ref is None:
        ref = y.clone()
    assert torch.equal(y, ref), "the reserved-CU launch must give the same bits"
    return dt


if __name__ == "__main__":
    print(f"stand-in collective: {W} workgroups x {D:.0f} us on a second stream, mode {MODE}", flush=True)
    for rep in range(int(os.environ.get("REPS", "2"))):
        base = run(0, False)
        print(f"  no collective, all 256 CUs: {base:.4f} ms per step", flush=True)
        for n in (0, 8, 16, 24, 32):
            if n > 0:
                print(f"  reserve {n:2d} CUs, no collective: {run(n, False):.4f} ms per step", flush=True)
            print(f"  reserve {n:2d} CUs, with the stand-in: {run(n, True):.4f} ms per step", flush=True)
