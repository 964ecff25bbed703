"""Exercise the RCCL code path of diffsptk_amd.dist on ONE GPU (a world-size-1 "nccl" process group with the
collectives forced on): catches API / layout / stream misuse of the overlapped in-place all-gather before the
8-GPU run, which only the driver can launch.  Run by tests/test_gpu_configs.py::test_rccl_path_world1."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402
from diffsptk_amd import dist as D  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
x = torch.randn(64, 16000, device=dev)
ref = mcep(stft(x))


def compute(w):
    return mcep(stft(w))


for nch in (1, 2, 4, 5):
    out = D.analyze_chunked_overlap(x, compute, nch, force_collective=True)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), f"rank-major gather mismatch at {nch} chunks"
cm = D.analyze_chunked_overlap(x, compute, 4, layout="chunk_major", force_collective=True)
torch.cuda.synchronize()
assert cm.shape == (4, 1, 16, 200, 25) and torch.equal(cm.reshape(64, 200, 25), ref)
# streaming use (bench.py, N > 1): one chunk, completion deferred behind the next batch
a, ha = D.analyze_chunked_overlap(x, compute, 1, defer=True, force_collective=True)
b, hb = D.analyze_chunked_overlap(x, compute, 1, defer=True, force_collective=True)
ha.wait()
hb.wait()
torch.cuda.synchronize()
assert torch.equal(a, ref) and torch.equal(b, ref)
g = D.all_gather_features(ref, 64)
big = ref.new_empty((64, 200, 25))
dist.all_gather_into_tensor(big, ref)
torch.cuda.synchronize()
assert torch.equal(big, ref) and torch.equal(g, ref)
dist.barrier()
dist.destroy_process_group()
print("nccl smoke OK")
