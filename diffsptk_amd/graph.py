"""HIP graphs for launch-bound calls.

Every operator of this package is a plain asynchronous launch on the current stream with caller-owned workspaces (no host
synchronisation, no library-side allocation), so a whole forward call -- STFT -> mcep is 2 launches, the mel-generalized
analysis ~45, the multi-stage MLSA filter ~25 -- can be captured once in a HIP graph and replayed on new data.  At small
batches those calls are bound by the ~10-30 us each launch costs from Python, not by the kernels.
"""
from __future__ import annotations

import torch


class Graphed:
    """`fn(*tensors)` captured in a HIP graph (torch.cuda.CUDAGraph) without an autograd graph.

    g = Graphed(lambda x: mcep(stft(x)), example_x)     # warm-up calls + capture on a side stream
    y = g(new_x)                                         # copies new_x into the static input, replays, returns the static outputs

    Shapes, dtypes and devices are fixed by the examples; the returned tensors are the graph's static outputs (overwritten
    by the next call: clone what must be kept).  Modules must be fully constructed (tables on the device) before the capture.
    """

    def __init__(self, fn, *examples: torch.Tensor, warmup: int = 2) -> None:
        if not examples or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in examples):
            raise ValueError("Graphed: the example inputs must be tensors on a HIP device")
        self._static_in = [t.detach().clone() for t in examples]
        self._graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=examples[0].device)
        side.wait_stream(torch.cuda.current_stream(examples[0].device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):   # operand images, LDS attributes, per-stream scratch, allocator
                fn(*self._static_in)
        torch.cuda.current_stream(examples[0].device).wait_stream(side)
        with torch.no_grad(), torch.cuda.graph(self._graph, stream=side):
            self._static_out = fn(*self._static_in)

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self._static_in):
            raise ValueError(f"Graphed: expected {len(self._static_in)} inputs")
        for dst, src in zip(self._static_in, inputs):
            if src.shape != dst.shape or src.dtype != dst.dtype or src.device != dst.device:
                raise ValueError(f"Graphed: input {tuple(src.shape)} {src.dtype} does not match the captured {tuple(dst.shape)} {dst.dtype}")
            dst.copy_(src)
        self._graph.replay()
        return self._static_out
