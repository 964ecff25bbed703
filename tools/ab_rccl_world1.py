import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import diffsptk_amd as dsp
from diffsptk_amd import ops
from diffsptk_amd.dist import analyze_chunked_overlap
x = torch.randn(1024, 16000, device=dev)
stft = dsp.STFT(400, 80, 512, device=dev); mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
fused = dsp.fuse(stft, mcep)
def run(depth, reserve, steps=200):
    os.environ["DSA_RESERVE_CUS"] = str(reserve)
    pend = []
    with torch.no_grad():
        def step():
            out, h = analyze_chunked_overlap(x, fused, 1, defer=True, force_collective=True)
            pend.append((out, h))
            while len(pend) > depth: pend.pop(0)[1].wait()
        for _ in range(30): step()
        while pend: pend.pop(0)[1].wait()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): step()
        while pend: pend.pop(0)[1].wait()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
with torch.no_grad():
    for _ in range(30): fused(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fused(x)
    torch.cuda.synchronize(); print(f"analysis alone: {(time.perf_counter()-t0)/200*1e3:.4f} ms per step")
for depth in (1, 2):
    for reserve in (0, 8):
        print(f"RCCL world of 1, all_gather_into_tensor of (1024, 200, 25) in place, completed {depth} step(s) later, {reserve} CUs reserved: {run(depth, reserve):.4f} ms per step", flush=True)
dist.destroy_process_group()
