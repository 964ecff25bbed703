import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import diffsptk_amd as dsp
from diffsptk_amd import ops, _lib
dev = "cuda"
x = torch.randn(1024, 16000, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
stft = dsp.STFT(400, 80, 512, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
def f_mod(): return mcep(stft(x))
def f_ops():
    X = ops.StftFn.apply(x, stft.window, stft.twiddle, 400, 80, 512, True, False, "constant", 1e-9, None, 3, 0)
    return ops.McepFn.apply(X, mcep.G, mcep.D, mcep.E, mcep.alpha_vector, 512, 24, 10, 0)
def run(f, K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(K): out = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
for f in (f_mod, f_ops): run(f, 5)
for name, f in (("module", f_mod), ("ops", f_ops), ("module", f_mod), ("ops", f_ops)):
    print(name, "K=20: %.4f" % run(f, 20), "K=40: %.4f" % run(f, 40), "K=200: %.4f" % run(f, 200))

# --- HIP events between the launches: torch's timing events vs raw events without the system-scope fence
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
hip.hipEventSynchronize.argtypes = [C.c_void_p]

def raw_event(flags):
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), flags) == 0
    return e

def f_torch_events():
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    X = ops.StftFn.apply(x, stft.window, stft.twiddle, 400, 80, 512, True, False, "constant", 1e-9, None, 3, 0)
    e[1].record()
    y = ops.McepFn.apply(X, mcep.G, mcep.D, mcep.E, mcep.alpha_vector, 512, 24, 10, 0)
    e[2].record()
    return y

def make_raw(flags):
    log = []
    def f():
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        e = [raw_event(flags) for _ in range(3)]
        hip.hipEventRecord(e[0], st)
        X = ops.StftFn.apply(x, stft.window, stft.twiddle, 400, 80, 512, True, False, "constant", 1e-9, None, 3, 0)
        hip.hipEventRecord(e[1], st)
        y = ops.McepFn.apply(X, mcep.G, mcep.D, mcep.E, mcep.alpha_vector, 512, 24, 10, 0)
        hip.hipEventRecord(e[2], st)
        log.append(e)
        return y
    return f, log

print("torch events every step  K=40: %.4f" % run(f_torch_events, 40))
for name, flags in (("raw default", 0), ("raw DisableSystemFence", 0x20000000), ("raw ReleaseToDevice", 0x40000000)):
    f, log = make_raw(flags)
    run(f, 3); log.clear()
    t = run(f, 40)
    ms = C.c_float()
    a = b = 0.0
    for e in log:
        hip.hipEventElapsedTime(C.byref(ms), e[0], e[1]); a += ms.value
        hip.hipEventElapsedTime(C.byref(ms), e[1], e[2]); b += ms.value
    print("%-24s K=40: %.4f ms/step | stft %.4f mcep %.4f" % (name, t, a / len(log), b / len(log)))
