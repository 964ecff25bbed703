"""Chunked STFT->mcep on one GPU with 1 or 2 compute streams (dev tool; models the compute side of
dist.analyze_chunked_overlap without the collective)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402

B = 1024
x = torch.randn(B, 16000, device="cuda")
stft = dsp.STFT(400, 80, 512).to("cuda")
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10).to("cuda")
f = lambda w: mcep(stft(w))
side = torch.cuda.Stream()


def run(n_chunks, n_streams):
    main = torch.cuda.current_stream()
    outs = []
    if n_streams > 1:
        side.wait_stream(main)
    for c in range(n_chunks):
        lo, hi = c * B // n_chunks, (c + 1) * B // n_chunks
        s = main if (n_streams == 1 or c % 2 == 0) else side
        with torch.cuda.stream(s):
            outs.append(f(x[lo:hi]))
    if n_streams > 1:
        main.wait_stream(side)
    return outs


for n_chunks, n_streams in ((1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (8, 2)):
    for _ in range(3):
        run(n_chunks, n_streams)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(n_chunks, n_streams)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"chunks {n_chunks} streams {n_streams}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f} ms")
