"""Does the packed STFT kernel's launch time depend on WHERE its buffers live?  Same launch (1024 utterances x 1 s, 204 800 frames) with the output /
input carved out of one big pool at different byte offsets, and after different amounts of unrelated allocation (tools/time_fused_mcep.py measures
68-69 us where bench.py's process measures 75-76 us for the same call on the same box)."""
import os, sys, statistics, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diffsptk_amd as dsp
from diffsptk_amd import _lib, ops
dev = "cuda"
B, T = 1024, 16000
stft = dsp.STFT(400, 80, 512, device=dev)
N, K = 200, 257
def launch(xc, y):
    ops._call("dsa_stft_fwd", ops._p(xc), B, T, 400, 80, 512, ops._p(stft.window), ops._p(stft.twiddle), 1, 0, 0, 1e-9, 0, 0.0, 3,
              ops._dtype_code(xc), _lib.ALGO_AUTO, ops._p(y), ops._stream())
def timeit(fn, n=50, groups=5):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(out)
import time
x0 = torch.randn(B, T, device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5: stft(x0)
torch.cuda.synchronize()
print(f"module call: {timeit(lambda: stft(x0)):.1f} us; x at {x0.data_ptr():#x}")
pool = torch.empty(B * N * K + (64 << 20) // 4, device=dev)
xpool = torch.empty(B * T + (8 << 20) // 4, device=dev)
xpool[: B * T] = x0.view(-1)
for yoff in (0, 64, 256, 1024, 4096, 65536, 1 << 20, (1 << 21) + 4096, 37 << 20):
    y = pool[yoff // 4: yoff // 4 + B * N * K].view(B, N, K)
    print(f"y at pool + {yoff:>9}: ({y.data_ptr():#x}) {timeit(lambda: launch(x0, y)):.1f} us")
y = pool[: B * N * K].view(B, N, K)
for xoff in (0, 16, 256, 4096, 1 << 20):
    xs = xpool[xoff // 4: xoff // 4 + B * T].view(B, T)
    xs.copy_(x0)
    print(f"x at pool + {xoff:>9}: ({xs.data_ptr():#x}) {timeit(lambda: launch(xs, y)):.1f} us")
# unrelated allocations in between (what bench.py's process has done by the time it measures this launch)
junk = [torch.empty(sz, device=dev) for sz in (3 << 20, 50 << 20, 7 << 20, 225 << 18, 626 << 18)]
print(f"module call after {sum(j.numel() for j in junk) * 4 >> 20} MB of other allocations: {timeit(lambda: stft(x0)):.1f} us")
del junk
ys = [torch.empty(B, N, K, device=dev) for _ in range(3)]
for i, yy in enumerate(ys):
    print(f"fresh torch.empty output {i} ({yy.data_ptr():#x}): {timeit(lambda: launch(x0, yy)):.1f} us")
