"""Exercise the RCCL code path of diffsptk_amd.dist on ONE GPU (world_size 1 process group):
catches API / stream misuse of the chunked overlapped all-gather before the 8-GPU run."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp
from diffsptk_amd import dist as D

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
x = torch.randn(64, 16000, device=dev)
ref = mcep(stft(x))
# bypass the world == 1 shortcut to run the collectives for real
world = dist.get_world_size()
out = None; pending = []
for lo, hi in [D.shard_bounds(64, 4, c) for c in range(4)]:
    feat = mcep(stft(x[lo:hi])).contiguous()
    if out is None: out = feat.new_empty((world, 64, *feat.shape[1:]))
    pending.append((dist.all_gather([out[r, lo:hi] for r in range(world)], feat, async_op=True), feat))
for w, _ in pending: w.wait()
torch.cuda.synchronize()
assert torch.equal(out.reshape(64, *out.shape[2:]), ref), "chunked gather mismatch"
g = D.all_gather_features(ref, 64)
big = ref.new_empty((64, 200, 25)); dist.all_gather_into_tensor(big, ref); torch.cuda.synchronize()
assert torch.equal(big, ref) and torch.equal(g, ref)
dist.barrier(); dist.destroy_process_group()
print("nccl smoke OK")
