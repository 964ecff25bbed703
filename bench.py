#!/usr/bin/env python3
"""Headline benchmark: frames/sec of STFT -> mel-cepstrum (fl=400 fp=80 nfft=512 M=24,
alpha=0.42, n_iter=10) on synthetic 16 kHz waveforms, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]           (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over this rank's shard of utterances (inputs resident in
HBM): fused STFT kernel -> mel-cepstral analysis kernel -> (N > 1) ONE in-place all-gather of the
(B, 200, 25) features over RCCL, left in flight and completed two steps later (dist.GATHER_DEPTH) so that it runs behind the
next step's kernels.  Weak scaling: 1024 utterances x 1 s per GPU, so N = 8 is BASELINE.json configs[4]
(batch 8192 sharded 8x) and N = 1 is its per-GPU shard.

Timing: an untimed clock ramp (--ramp-seconds, default 0.3 s of the same steps: the first ~20 ms after idle run
10 % slower), W warmup steps, then exactly K steps between barrier + synchronize; defaults K = 200, W = 20
(0.2 s of GPU time).  HIP events bracket the two launches of every fourth timed step (per-kernel averages).

Rank 0 prints ONE JSON line (contract in the task statement).  Besides the headline value it carries
  roofline        the dominant kernel (mel-cepstral analysis): bound "valu_issue" -- vector wave-instructions
                  issued per second against the machine's issue rate -- with the algorithmic-flop figure beside it;
  roofline_stft   the fused Frame+Window+rFFT stage (HBM bound, 1348 B/frame): the north star's ">= 50 %" number;
  configs         the other BASELINE configs measured in the same run (N = 1): config 2 (STFT, batch 64, and the
                  batch sweep), config 3 (STFT+mcep forward+backward, batch 256 and 1024, with roofline_bwd),
                  config 4 (fused Frame+Window+LPC, batch 1024, with its roofline);
  cpu_baseline    the reference's op sequence with stock PyTorch CPU ops (oracle/torch_port.py) on the host cores,
                  at 1 / 8 / 32 / all physical cores (value = the best), with the host description (N = 1 only).
`traffic` / `pmc` fields are STATIC: rocprofv3 --pmc passes of this command, committed under profiles/ (a counter
pass cannot be taken from inside the process being timed); every other number is measured live in this run.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FL, FP, NFFT, M, ALPHA, N_ITER = 400, 80, 512, 24, 0.42, 10
SAMPLES = 16000
FRAMES_PER_UTT = (SAMPLES - 1) // FP + 1  # 200

# algorithmic work per frame
STFT_BYTES_PER_FRAME = FP * 4 + (NFFT // 2 + 1) * 4  # 320 B read + 1028 B written = 1348 B
STFT_BWD_BYTES_PER_FRAME = (NFFT // 2 + 1) * 4 + FP * 4 + FP * 4  # grad 1028 B + waveform 320 B read, 320 B written
K, M1, M2 = NFFT // 2 + 1, M + 1, 2 * M + 1
_SOLVE_MAC = M1 ** 3 // 3 + M1 * M1  # Cholesky-class elimination + substitutions
MCEP_FLOP_PER_FRAME = 2 * (K * M1 + N_ITER * (M1 * K + K * M2 + _SOLVE_MAC)) + (N_ITER + 1) * K
# backward of the unrolled iteration, per step: both forward chains again, one elimination with two right-hand
# sides, rtbar (Toeplitz + Hankel diagonals), ebar = E rtbar, mbar -= 2 D zbar; once: lbar += G mbar_0
MCEP_BWD_FLOP_PER_FRAME = 2 * (N_ITER * (M1 * K + K * M2 + M1 ** 3 // 3 + 2 * M1 * M1 + 2 * M1 * M1 + K * M2 + K * M1) + K * M1) \
    + N_ITER * 3 * K
# the binding pipe's ALGORITHMIC bound for the step (round-4 review): work that has to run on the float32 datapath -- the ten
# eliminations, the exponentials, log and (one-launch step) the 512-point real FFT: ~2.5 n log2 n -- priced at the float32 peak; the
# three matrix chains run as 3-term binary16 splits on the separate matrix pipe, priced at the dense binary16 peak.  The larger of
# the two times over the measured launch time is `roofline.algorithmic_frac`: an efficiency (not the issued-instruction
# utilisation that `frac` reports).
MCEP_F32_FLOP_PER_FRAME = N_ITER * 2 * _SOLVE_MAC + (N_ITER + 1) * K
STFT_F32_FLOP_PER_FRAME = int(2.5 * NFFT * 9)
MCEP_F16_FLOP_PER_FRAME = 3 * 2 * (K * M1 + N_ITER * (M1 * K + K * M2))
F16_MFMA_PEAK_TFLOPS = 2500.0   # dense binary16 matrix peak (MI355X_MICROARCH.md)
LPC_FLOP_PER_FRAME = 2 * FL * M1 + 2 * M * M + 4 * M   # direct lag sums + Levinson recursion, float64
LPC_BYTES_PER_FRAME = FP * 4 + M1 * 4
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3    # fp32 vector peak with packed instructions = fp32 MFMA dense peak (same guide)
FP64_PEAK_TFLOPS = 78.6     # fp64 vector peak (same guide)
VALU_ISSUE_PEAK_GIPS = 1024 * 2.4 / 2   # 1024 SIMD-32s, one wave64 vector instruction per 2 cycles (MI355X_MICROARCH.md), 2.4 GHz: 1228.8 G wave-instr/s
# The unit that binds the mel-cepstral kernels is the float32 datapath of a SIMD, shared by the vector ALU and the float32
# matrix instructions.  Measured occupancy of that datapath per wave64 instruction (rocprofv3 --pmc on tools/bench_issue.cpp,
# profiles/r03_issue_calibration.txt): multiply-add class / DPP / conversions / packed 4 cycles, two-operand and move class 2,
# transcendentals 8, v_mfma_f32_4x4x1 8.  Peak = every SIMD busy every cycle at the nominal clock.
F32_DATAPATH_PEAK_GCPS = 1024 * 2.4        # 2457.6 G datapath-cycles/s
ISA_MIX_FILE = os.path.join("profiles", "r03_isa_mix.json")


def isa_mix(kernel_substring: str):
    """Instruction mix of a kernel's Newton-step loop priced with the measured issue costs (tools/isa_mix.py): read off the
    loaded library when llvm-objdump is there, else the committed profiles/r03_isa_mix.json of the same source."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_mix as im
        from diffsptk_amd import _lib as L

        ks = im.disassemble(L.LIB_PATH)
        for sym, ins in ks.items():
            if kernel_substring in sym and ins:
                lp = im.innermost_last_loop(ins)
                c, cyc = im.mix(ins[lp[0]: lp[1] + 1])
                return {"f32_datapath_cycles_per_pass": sum(cyc.values()), "counts": dict(c), "frames_per_pass": 16,
                        "xdl_cycles_per_pass": 16 * c["mfma_f16_xdl"], "source": "live: llvm-objdump of " + os.path.relpath(L.LIB_PATH, ROOT)}
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, ISA_MIX_FILE)) as f:
            d = json.load(f)
        for sym, v in d.items():
            if kernel_substring in sym:
                v = dict(v)
                v["source"] = ISA_MIX_FILE + " (static)"
                return v
    except Exception:
        pass
    return None


def datapath_roofline(mixd, n_iter_steps, frames, seconds):
    """achieved / peak of the float32 datapath: the priced instruction mix of one Newton step over 16 frames x steps per frame
    x frames / launch time, against 1024 SIMDs x 2.4 GHz."""
    if not mixd:
        return None
    per_frame = mixd["f32_datapath_cycles_per_pass"] * n_iter_steps / mixd["frames_per_pass"]
    a = per_frame * frames / seconds / 1e9
    return {"achieved": a, "peak": F32_DATAPATH_PEAK_GCPS, "unit": "G datapath-cycles/s", "frac": a / F32_DATAPATH_PEAK_GCPS,
            "datapath_cycles_per_frame": per_frame, "mix_per_16_frames_and_step": mixd["counts"], "mix_source": mixd["source"],
            "xdl_cycles_per_frame": mixd["xdl_cycles_per_pass"] * n_iter_steps / mixd["frames_per_pass"]}


PMC_FILE = os.path.join("profiles", "pmc_traffic.json")


def pmc_static(kernel: str):
    """The committed rocprofv3 --pmc record of `kernel` (profiles/pmc_traffic.json: counters of THIS command taken
    in separate passes, tools/pmc_passes.sh), or None."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        k["_source"] = d.get("source", PMC_FILE)
        return k
    except Exception:
        return None


def grad_and_kernel(_lib, y, x, cot):
    """(gradient, name of the kernel family that produced it): dsa_last_kernel() is per thread and the backward runs on
    autograd's worker thread, so a hook on x reads it there."""
    names = []
    h = x.register_hook(lambda g: names.append(_lib.last_kernel()))
    (gx,) = torch.autograd.grad(y, x, cot, retain_graph=True)
    h.remove()
    return gx, (names[-1] if names else None)


def pmc_traffic(kernel: str, frames: float):
    """HBM bytes per launch of `kernel` from the static PMC record (FETCH_SIZE and WRITE_SIZE collected separately and
    corrected as MI355X_MICROARCH.md prescribes), rescaled to this launch's frame count; None without a record."""
    k = pmc_static(kernel)
    return None if k is None else k["hbm_bytes_per_launch"] * frames / k["frames_per_launch"]


def host_description():
    d = {"logical_cpus": os.cpu_count(), "torch": torch.__version__}
    try:
        import psutil

        d["physical_cores"] = psutil.cpu_count(logical=False)
    except Exception:
        d["physical_cores"] = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                d["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    cfg = torch.__config__.show()
    d["blas"] = next((t.strip() for t in cfg.replace("\n", ",").split(",") if "BLAS_INFO" in t), None)
    d["mkl"] = next((ln.strip(" -") for ln in cfg.split("\n") if "Math Kernel Library" in ln or "oneAPI Math" in ln), None)
    d["mkldnn"] = next((ln.strip(" -") for ln in cfg.split("\n") if "MKL-DNN" in ln), None)
    d["openmp"] = next((ln.strip(" -") for ln in cfg.split("\n") if "OpenMP" in ln), None)
    return d


def _cpu_time(fn, budget_s, min_runs, max_runs):
    fn()   # warm-up
    times, t_all = [], time.perf_counter()
    while len(times) < min_runs or (time.perf_counter() - t_all < budget_s and len(times) < max_runs):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return statistics.median(times), len(times)


def cpu_baseline():
    """the reference's op sequence with stock ATen CPU operators (oracle/torch_port.py) on the same
    synthetic input, float32, at several thread counts -- one core, 8 (the survey's container), 32 and all physical
    cores -- each a median of >= 10 runs of a 32- / 64-utterance sample; host described.  `value` is the BEST of them (on
    a many-core host the small ATen calls of this path lose more to threading than they gain: one core beats 128).
    ~30 s of CPU time in total."""
    from oracle import torch_port as TP

    host = host_description()
    n_all = host.get("physical_cores") or torch.get_num_threads()
    tab = TP.McepTables(NFFT, M, ALPHA, torch.float32)
    w = TP.window_table(FL, dtype=torch.float32)
    gen = torch.Generator().manual_seed(0)
    res = {}
    prev = torch.get_num_threads()
    plan = [(1, 32, 4.0), (8, 64, 4.0), (32, 64, 4.0), (n_all, 64, 4.0)]   # (threads, utterances, time budget beyond the 10 runs)
    seen = set()
    try:
        with torch.no_grad():
            for nthreads, B, budget in plan:
                nthreads = max(1, min(int(nthreads), int(n_all)))
                if nthreads in seen:
                    continue
                seen.add(nthreads)
                torch.set_num_threads(nthreads)
                x = torch.randn(B, SAMPLES, generator=gen)
                med, n = _cpu_time(lambda: TP.stft_mcep(x, tab, FL, FP, N_ITER, w), budget, 10, 12)   # median of >= 10 runs
                res[str(nthreads)] = {"value": B * FRAMES_PER_UTT / med, "threads": nthreads, "utterances": B,
                                      "frames": B * FRAMES_PER_UTT, "median_s": med, "runs": n}
    finally:
        torch.set_num_threads(prev)
    best = max(res.values(), key=lambda r: r["value"])
    return {
        "value": best["value"], "unit": "frames/s", "cores": int(n_all), "threads_best": best["threads"], "kind": "port",
        "sample": f"{best['utterances']} utterances x 1 s (={best['frames']} frames) STFT->mcep forward, float32, stock torch CPU "
                  f"ops (oracle/torch_port.py: the reference's ATen call sequence), best of {sorted(int(k) for k in res)} threads "
                  f"({best['threads']}; the host has {n_all} physical cores), median of {best['runs']} runs",
        "by_threads": res, "one_core": res.get("1"), "all_cores": res.get(str(max(int(k) for k in res))), "host": host,
    }


def c_oracle_baseline():
    """The C restatement (oracle/sptk_oracle.c, OpenMP over frames in mcep) on a small sample."""
    import numpy as np

    from oracle import oracle as O

    B = 8
    x = np.random.default_rng(0).standard_normal((B, SAMPLES)).astype(np.float32)
    O.stft_mcep(x[:1])
    t0 = time.perf_counter()
    O.stft_mcep(x)
    dt = time.perf_counter() - t0
    return {"value": B * FRAMES_PER_UTT / dt, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{B} utterances x 1 s, C oracle (naive mixed-radix FFT + LU), float32, single run"}


def gpu_time(fn, n=30, groups=3):
    """ms per call of fn(): `groups` groups of n back-to-back calls between two HIP events on the current stream,
    median over the groups (the launches of a group pipeline through the command processor: kernel time, not
    kernel + dispatch gap)."""
    # no cyclic garbage collection inside a timed group (as the standard library's timeit): a generation-2 pass of this process
    # takes 18 - 80 ms of host time and, landing between two launches, reads as a slow kernel (profiles/r05_measured_tolerances_final.txt).
    # Only disabled, never forced here: a forced collection idles the chip for tens of milliseconds right in front of the groups, and the
    # clock it costs showed in every figure (config 3's forward 0.61 -> 0.69 ms).
    gc_was = gc.isenabled()
    gc.disable()
    fn()
    torch.cuda.synchronize()
    out = []
    try:
        for _ in range(groups):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / n)
    finally:
        if gc_was:
            gc.enable()
    return statistics.median(out)


def other_configs(dsp, ops, _lib, dev, stft, mcep, x1024):
    """BASELINE configs 2, 3, 4 in the same process (N = 1), each with the roofline that bounds it."""
    res = {}
    # ---- config 2: fused Frame+Window+rFFT kernel, batch 64 (and the batch sweep: where the stage reaches 50 %) ----
    sweep = {}
    with torch.no_grad():
        for B in (64, 256, 1024, 4096):
            xb = x1024[:B] if B <= x1024.size(0) else torch.randn(B, SAMPLES, device=dev)
            t = gpu_time(lambda: stft(xb), n=40) * 1e-3
            fr = B * FRAMES_PER_UTT
            sweep[str(B)] = {"us_per_launch": t * 1e6, "GB/s": STFT_BYTES_PER_FRAME * fr / t / 1e9,
                             "frac": STFT_BYTES_PER_FRAME * fr / t / 1e9 / HBM_PEAK_GBS}
            del xb
    c2 = sweep["64"]
    res["config2_stft_batch64"] = {
        "workload": "BASELINE configs[1]: fused Frame+Window+rFFT kernel, 64 utterances x 1 s (12 800 frames, 17.3 MB)",
        "frames/s": 64 * FRAMES_PER_UTT / (c2["us_per_launch"] * 1e-6),
        "roofline": {"kernel": "stft512_fwd", "bound": "hbm", "achieved": c2["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": c2["frac"], "traffic": None, "avg_launch_ms": c2["us_per_launch"] * 1e-3,
                     "note": "one pass per wave: the launch is latency-bound (launch + table prologue + one memory round trip "
                             "+ the end-of-kernel write-back), 17 MB cannot amortise it; see batch_sweep for the size "
                             "at which the stage reaches its large-batch rate"},
        "batch_sweep": sweep, "timing": "back-to-back launches (gpu_time)",
    }
    # ---- config 3: STFT + mcep forward + backward ----
    for B in (256, 1024):
        xb = x1024[:B]
        fr = B * FRAMES_PER_UTT

        def fwdbwd():
            xg = xb.clone().requires_grad_(True)
            mcep(stft(xg)).mean().backward()
            return xg.grad

        t_fb = gpu_time(fwdbwd, n=10) * 1e-3
        with torch.no_grad():
            t_f = gpu_time(lambda: mcep(stft(xb)), n=10) * 1e-3
        xg = xb.clone().requires_grad_(True)
        X = stft(xg)
        Xd = X.detach().requires_grad_(True)
        mc = mcep(Xd)
        g = torch.ones_like(mc) / mc.numel()
        t_mb = gpu_time(lambda: torch.autograd.grad(mc, Xd, g, retain_graph=True), n=10) * 1e-3
        gX, k_bwd = grad_and_kernel(_lib, mc, Xd, g)
        t_sb = gpu_time(lambda: torch.autograd.grad(X, xg, gX, retain_graph=True), n=10) * 1e-3
        _, k_sbwd = grad_and_kernel(_lib, X, xg, gX)
        pm = pmc_static("mcep_mfma_bwd")
        ipf = pm["derived"]["valu_insts_per_frame"] if pm else None
        res[f"config3_fwdbwd_batch{B}"] = {
            "workload": f"BASELINE configs[2]: STFT + mcep forward + backward of mean(), {B} utterances x 1 s ({fr} frames)",
            "ms_fwd": t_f * 1e3, "ms_fwd_bwd": t_fb * 1e3, "frames/s": fr / t_fb,
            "ms_mcep_bwd": t_mb * 1e3, "ms_stft_bwd": t_sb * 1e3,
            "roofline_bwd": {
                "kernel": "mcep_mfma_bwd", "bound": "valu_issue",
                "achieved": (ipf * fr / t_mb / 1e9) if ipf else None, "peak": VALU_ISSUE_PEAK_GIPS,
                "unit": "G wave-instr/s", "frac": (ipf * fr / t_mb / 1e9 / VALU_ISSUE_PEAK_GIPS) if ipf else None,
                "traffic": pmc_traffic("mcep_mfma_bwd", fr), "avg_launch_ms": t_mb * 1e3,
                "algorithmic": {"achieved": MCEP_BWD_FLOP_PER_FRAME * fr / t_mb / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": MCEP_BWD_FLOP_PER_FRAME * fr / t_mb / 1e12 / FP32_PEAK_TFLOPS,
                                "flop_per_frame": MCEP_BWD_FLOP_PER_FRAME},
                "datapath": datapath_roofline(isa_mix("mcep_mfma_bwd2_kernel_h"), N_ITER, fr, t_mb),
                "pmc": pm["derived"] if pm else None, "pmc_source": pm["_source"] if pm else None,
                "arith": "three matrix chains as 3-term binary16 MFMA splits (fp32 accumulate; the forward's rt rows are saved), "
                         "two-right-hand-side 25x25 block elimination + the u g^T outer product on v_mfma_f32_4x4x1, two waves per SIMD", "last_kernel": k_bwd,
            },
            "roofline_stft_bwd": (lambda pmb: {
                "kernel": k_sbwd, "bound": "hbm", "achieved": STFT_BWD_BYTES_PER_FRAME * fr / t_sb / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": STFT_BWD_BYTES_PER_FRAME * fr / t_sb / 1e9 / HBM_PEAK_GBS,
                "traffic": pmc_traffic("stft512_bwd", fr), "avg_launch_ms": t_sb * 1e3, "bytes_per_frame": STFT_BWD_BYTES_PER_FRAME,
                "pmc": pmb["derived"] if pmb else None, "pmc_source": pmb["_source"] if pmb else None,
                "note": "one launch: forward transform recomputed, cotangent packed, inverse transform, overlap-add carried in "
                        "registers (csrc/stft_bwd_pk.h); 1668 algorithmic bytes per frame.  Not HBM-bound in practice: the LDS "
                        "pipe (two transposes per transform, four transforms' worth per pass) and vector issue share the time "
                        "(pmc: lds_busy / valu_busy), DESIGN.md 3.1"})(pmc_static("stft512_bwd")),
            "timing": "back-to-back calls through the module API (gpu_time)",
        }
        del xg, X, Xd, mc, g, gX
    # ---- config 4: LPC branch, fused Frame + Window + acorr + Levinson ----
    w = dsp.Window(FL, device=dev).window
    B = 1024
    fr = B * FRAMES_PER_UTT
    # (the entry timed is the drop-in one: fuse(frame, window, lpc) over the reference's own modules -- README.md:198-201 of the
    #  reference -- one launch forward, one launch backward; the module chain beside it)
    frm, wn, lpc = dsp.Frame(FL, FP), dsp.Window(FL, device=dev), dsp.LPC(FL, M, eps=1e-5, device=dev)
    flpc = dsp.fuse(frm, wn, lpc)
    with torch.no_grad():
        t_l = gpu_time(lambda: flpc(x1024), n=30) * 1e-3
        k_lpc = _lib.last_kernel()
        t_chain = gpu_time(lambda: lpc(wn(frm(x1024))), n=10) * 1e-3

    def lpc_fb():
        xg = x1024.clone().requires_grad_(True)
        lpc(wn(frm(xg))).mean().backward()

    def flpc_fb():
        xg = x1024.clone().requires_grad_(True)
        flpc(xg).mean().backward()

    t_lfb = gpu_time(lpc_fb, n=10) * 1e-3
    t_ffb = gpu_time(flpc_fb, n=10) * 1e-3
    gl_ = torch.randn(B, FRAMES_PER_UTT, M + 1, device=dev)
    gxl_ = torch.empty_like(x1024)
    t_lb = gpu_time(lambda: ops._call("dsa_frame_window_lpc_bwd", gl_.data_ptr(), x1024.data_ptr(), B, x1024.size(-1), FL, FP, w.data_ptr(), 1, 0, M,
                                      1e-5, _lib.F32, gxl_.data_ptr(), ops._stream()), n=30) * 1e-3
    del gl_, gxl_
    res["config4_lpc_batch1024"] = {
        "workload": f"BASELINE configs[3]: Frame+Window+acorr+levdur (M=24), {B} utterances x 1 s ({fr} frames), one wave per 64 frames",
        "frames/s": fr / t_l, "ms_fused_fwd": t_l * 1e3, "ms_fused_fwd_bwd": t_ffb * 1e3, "ms_fused_bwd_launch": t_lb * 1e3,
        "entry": "diffsptk_amd.fuse(Frame, Window, LPC) (path: %s)" % flpc.last_path,
        "ms_module_chain_fwd": t_chain * 1e3, "ms_module_chain_fwd_bwd": t_lfb * 1e3,
        "roofline": (lambda mm: {
                     "kernel": k_lpc, "bound": "valu_issue (binary16-split Gram products on the matrix pipe; float64 lag sums and Levinson on the vector ALU)" if mm
                     else "valu_issue (float64)",
                     # executed matrix flops: 21 v_mfma_f32_16x16x4_f32 per frame (2048 flops each) against the dense float32 matrix peak;
                     # exact kernel: algorithmic float64 flops against the float64 vector peak
                     # the binding unit is the vector ALU (window, binary16 split, float64 lag sums and Levinson): vector wave-instructions per
                     # frame (static: profiles/) against the nominal issue rate; exact kernel: float64 flops against the float64 vector peak
                     "achieved": ((pmc_static(k_lpc) or {}).get("derived", {}).get("valu_insts_per_frame", 0) * fr / t_l / 1e9) if mm
                     else LPC_FLOP_PER_FRAME * fr / t_l / 1e12,
                     "peak": VALU_ISSUE_PEAK_GIPS if mm else FP64_PEAK_TFLOPS, "unit": "G wave-instr/s" if mm else "TFLOP/s (fp64 vector)",
                     "frac": ((pmc_static(k_lpc) or {}).get("derived", {}).get("valu_insts_per_frame", 0) * fr / t_l / 1e9 / VALU_ISSUE_PEAK_GIPS) if mm
                     else LPC_FLOP_PER_FRAME * fr / t_l / 1e12 / FP64_PEAK_TFLOPS,
                     "traffic": pmc_traffic(k_lpc, fr), "avg_launch_ms": t_l * 1e3, "flop_per_frame": LPC_FLOP_PER_FRAME,
                     "algorithmic_frac_f32": LPC_FLOP_PER_FRAME * fr / t_l / 1e12 / FP32_PEAK_TFLOPS,
                     "pmc": (pmc_static(k_lpc) or {}).get("derived"),
                     "pmc_source": (pmc_static(k_lpc) or {}).get("_source"),
                     "hbm": {"achieved": LPC_BYTES_PER_FRAME * fr / t_l / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": LPC_BYTES_PER_FRAME * fr / t_l / 1e9 / HBM_PEAK_GBS},
                     "note": "420 B/frame: far from the HBM roof.  Round 4: lag sums as a banded Gram product on v_mfma_f32_16x16x32_f16 (9 per frame, "
                             "3-term binary16 splits; float64 only to add the 16 entries of a lag and for the recursion), DESIGN.md 3.3; "
                             "DSA_LPC_LAGSUMS=f64 selects the exact float64-vector kernel of rounds 1-3"})("mfma" in (k_lpc or "")),
        "timing": "back-to-back launches (gpu_time)",
    }
    # ---- STFT -> mel filter bank, the fused launch against the two kernels ----
    fb = dsp.MelFilterBankAnalysis(fft_length=NFFT, n_channel=40, sample_rate=16000, use_power=True, device=dev)
    fused = dsp.fuse(stft, fb)
    with torch.no_grad():
        fused(x1024)
        k_f = _lib.last_kernel()
        t_fu = gpu_time(lambda: fused(x1024), n=30) * 1e-3
        t_2 = gpu_time(lambda: fb(stft(x1024)), n=30) * 1e-3
    def fb_fwdbwd(mod):
        xg = x1024.detach().requires_grad_(True)
        y = mod(xg)
        y.backward(torch.ones_like(y))

    t_fu_fb = gpu_time(lambda: fb_fwdbwd(fused), n=10) * 1e-3
    path_fb = fused.last_path
    t_2_fb = gpu_time(lambda: fb_fwdbwd(lambda t: fb(stft(t))), n=10) * 1e-3
    pm = pmc_static("stft512_fbank_fwd")
    ipf = pm["derived"]["valu_insts_per_frame"] if pm else None
    fb_bytes = 320 + 4 * 40
    res["f1_stft_fbank_batch1024"] = {
        "workload": f"SURVEY 8(f) row 1: STFT -> MelFilterBankAnalysis (40 channels, power domain), {B} utterances x 1 s ({fr} frames), "
                    "one launch (diffsptk_amd.fuse): the (B, N, 257) spectrogram never reaches memory",
        "path": fused.last_path, "frames/s": fr / t_fu, "ms_fused": t_fu * 1e3, "ms_two_kernels": t_2 * 1e3,
        "fwd_bwd": {"ms_fused": t_fu_fb * 1e3, "ms_two_stage": t_2_fb * 1e3, "path": path_fb,
                    "note": "forward + backward of sum(y) w.r.t. the waveform through the module API: fused = the one-launch forward, "
                            "dsa_fbank_bins_bwd (channel cotangents -> bins) and the packed STFT backward, no spectrogram kept; "
                            "two-stage = STFT and filter bank as separate differentiable modules"},
        "roofline": {"kernel": k_f, "bound": "valu_issue",
                     "achieved": (ipf * fr / t_fu / 1e9) if ipf else None, "peak": VALU_ISSUE_PEAK_GIPS, "unit": "G wave-instr/s",
                     "frac": (ipf * fr / t_fu / 1e9 / VALU_ISSUE_PEAK_GIPS) if ipf else None,
                     "traffic": pmc_traffic("stft512_fbank_fwd", fr), "avg_launch_ms": t_fu * 1e3,
                     "hbm": {"achieved": fb_bytes * fr / t_fu / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fb_bytes * fr / t_fu / 1e9 / HBM_PEAK_GBS, "bytes_per_frame": fb_bytes},
                     "pmc": pm["derived"] if pm else None, "pmc_source": pm["_source"] if pm else None,
                     "note": "480 algorithmic bytes per frame (320 in, 160 out): 12 us at the HBM peak -- with the spectrogram's "
                             "1028 B/frame round trip gone the stage is bound by vector issue and LDS cycles (FFT butterflies + "
                             "the segmented scans of the filter bank), DESIGN.md 3.35"},
        "timing": "back-to-back launches (gpu_time)",
    }
    # ---- one timing per module at 256 utterances x 1 s (51 200 frames) ----
    Bq = 256
    xq = x1024[:Bq]
    frq = Bq * FRAMES_PER_UTT
    rows = {}

    def timed(name, fn, n=5, bytes_per_frame=None, flop_per_frame=None):
        """`roofline`: the module's algorithmic bytes per frame (its inputs + outputs once, float32) against the HBM peak, and for
        the rows that are arithmetic by construction their multiply-adds against the float32 vector peak."""
        rows[name] = {"ms": gpu_time(fn, n=n, groups=2), "kernel": _lib.last_kernel()}
        t_s = rows[name]["ms"] * 1e-3
        rows[name]["Mframes/s"] = frq / rows[name]["ms"] / 1e3
        # ONE `roofline` per row, against the unit that binds it: rows whose arithmetic takes longer at its peak than their bytes at
        # the HBM peak are priced as arithmetic (float32 matrix / packed-vector peak), the others as memory; the other figure rides
        # along as `roofline_other`
        hbm = arith = None
        if bytes_per_frame is not None:
            a = bytes_per_frame * frq / t_s / 1e9
            hbm = {"bound": "hbm", "bytes_per_frame": bytes_per_frame, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": a / HBM_PEAK_GBS}
        if flop_per_frame is not None:
            a = flop_per_frame * frq / t_s / 1e12
            arith = {"bound": "f32 arithmetic (matrix / packed-vector peak)", "flop_per_frame": flop_per_frame, "achieved": a,
                     "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a / FP32_PEAK_TFLOPS}
        t_h = bytes_per_frame * frq / (HBM_PEAK_GBS * 1e9) if bytes_per_frame else 0.0
        t_a = flop_per_frame * frq / (FP32_PEAK_TFLOPS * 1e12) if flop_per_frame else 0.0
        first, other = (arith, hbm) if t_a > t_h else (hbm, arith)
        if first is not None:
            rows[name]["roofline"] = first
        if other is not None:
            rows[name]["roofline_other"] = other

    with torch.no_grad():
        Xq = stft(xq)
        stc = dsp.STFT(FL, FP, NFFT, out_format="complex", device=dev)
        ist = dsp.ISTFT(FL, FP, NFFT, device=dev)
        Zq = stc(xq)
        timed("f2 ISTFT", lambda: ist(Zq), n=10, bytes_per_frame=8 * (NFFT // 2 + 1) + 4 * FP)
        gl = dsp.GriffinLim(FL, FP, NFFT, n_iter=4, init_phase="zeros", device=dev)
        timed("f2 GriffinLim (4 iterations)", lambda: gl(Xq, out_length=xq.size(-1)), n=3,
              bytes_per_frame=4 * (NFFT // 2 + 1) + 4 * FP + 4 * (2 * (8 * (NFFT // 2 + 1) + 4 * FP)))   # in + out + 4 x (ISTFT, STFT)
        for it in (0, 3):
            ca = dsp.CepstralAnalysis(fft_length=NFFT, cep_order=M, n_iter=it, device=dev)
            # n_iter = 0: one (K x M1) product; each refinement step two more half-length transforms (5 N/2 log2(N/2) flops each)
            timed(f"f3 CepstralAnalysis n_iter={it}", lambda: ca(Xq), n=10, bytes_per_frame=4 * (NFFT // 2 + 1) + 4 * (M + 1),
                  flop_per_frame=2 * K * M1 + it * 2 * 5 * (NFFT // 2) * 8)
        mg = dsp.MelGeneralizedCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=ALPHA, gamma=-0.5, n_iter=N_ITER, device=dev)
        # per Newton step: seven row products (2 x 24 x 257 in, 257 x (24 + 47 + 47 + 25 + 25) out), ~25 flops per bin of spectrum
        # arithmetic, the order-24 solve; the gamma = -1 start and the conversion are one step's worth more
        timed("f3 MelGeneralizedCepstralAnalysis gamma=-0.5 n_iter=10", lambda: mg(Xq), n=2,
              bytes_per_frame=4 * (NFFT // 2 + 1) + 4 * (M + 1),
              flop_per_frame=(N_ITER + 1) * (2 * K * (2 * M + 3 * M + 2 * M1 + 2 * M) + 25 * K + 2 * (M ** 3 // 3 + M * M)))
        mcq = mcep(Xq)
        m2s = dsp.MelGeneralizedCepstrumToSpectrum(M, NFFT, alpha=ALPHA, device=dev)
        timed("f4 mgc2sp", lambda: m2s(mcq), n=10, bytes_per_frame=4 * (NFFT // 2 + 1) + 4 * (M + 1), flop_per_frame=2 * M1 * K)
        m2b = dsp.MelCepstrumToMLSADigitalFilterCoefficients(M, ALPHA, device=dev)
        timed("f4 mc2b", lambda: m2b(mcq), n=10, bytes_per_frame=8 * (M + 1))
        exc = torch.randn(Bq, SAMPLES, device=dev)
        for mode, kw in (("multi-stage", {}), ("single-stage", {}), ("freq-domain", dict(frame_length=FL, fft_length=NFFT))):
            ml = dsp.MLSA(M, FP, alpha=ALPHA, mode=mode, device=dev, **kw)
            # time-domain modes: two rows x taps x samples multiply-adds per frame (200 taps x 20 Taylor stages | 2000 taps)
            flop = {"multi-stage": 2 * 2 * 200 * 20 * FP, "single-stage": 2 * 2 * 2000 * FP}.get(mode)
            timed(f"f4 MLSA {mode}", lambda: ml(exc, mcq), n=2, bytes_per_frame=8 * FP + 4 * (M + 1), flop_per_frame=flop)
    res["f_rows_batch256"] = {"workload": f"SURVEY 8(f) rows 2-4, {Bq} utterances x 1 s ({frq} frames / {Bq * SAMPLES} samples), float32, "
                                          "module API, default options unless named", "rows": rows,
                              "timing": "back-to-back calls (gpu_time); `kernel` = the last kernel family the call dispatched to"}
    # ---- the 48 kHz set-ups of utils/public.py:22-104 (no tuned kernel: row products on the float32 matrix instruction, batched
    # Toeplitz-plus-Hankel solve): 64 utterances x 1 s, forward ----
    rows48 = {}
    with torch.no_grad():
        for B48 in (64, 512):   # 64 utterances: one workgroup per CU, latency-bound; 512: the chip filled (round-4 review: report both)
            x48 = torch.randn(B48, 48000, device=dev)
            for (fl, fp, nfft, m, a) in ((1200, 240, 2048, 49, 0.55), (1024, 256, 1024, 34, 0.55), (800, 200, 1024, 34, 0.55)):
                st48 = dsp.STFT(fl, fp, nfft, device=dev)
                mc48 = dsp.MelCepstralAnalysis(fft_length=nfft, cep_order=m, alpha=a, n_iter=N_ITER, device=dev)
                X48 = st48(x48)
                k48s = _lib.last_kernel()
                mc48(X48)
                k48 = _lib.last_kernel()
                fr48 = X48.shape[0] * X48.shape[1]
                k_, m1_ = nfft // 2 + 1, m + 1
                # per frame: mc0 = log X G, then per step S = mc D, rt = e E (float32 matrix products), the order-(M+1) elimination
                flop48 = 2 * k_ * m1_ + N_ITER * (2 * k_ * m1_ + 2 * k_ * (2 * m + 1) + 2 * (m1_ ** 3) // 3)
                t_s, t_m = gpu_time(lambda: st48(x48), n=5) * 1e-3, gpu_time(lambda: mc48(X48), n=5) * 1e-3
                by48 = 4 * fp + 4 * k_   # algorithmic bytes per frame of the STFT: frame_period new samples in, fft_length / 2 + 1 power values out
                rows48[f"fft {nfft} order {m}" + ("" if fl != 800 else " frame 800/200") + ("" if B48 == 64 else f" ({B48} utterances)")] = {
                    "utterances": B48, "frames": fr48, "stft_ms": t_s * 1e3, "mcep_ms": t_m * 1e3, "mcep_us_per_1000_frames": t_m * 1e9 / fr48,
                    "last_kernel": k48, "stft_kernel": k48s, "frames_per_s": fr48 / (t_s + t_m),
                    "roofline_stft": {"kernel": k48s, "bound": "hbm", "bytes_per_frame": by48, "achieved": by48 * fr48 / t_s / 1e9,
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by48 * fr48 / t_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                      "avg_launch_ms": t_s * 1e3},
                    "roofline": {"bound": "f32 arithmetic (matrix / packed-vector peak)", "flop_per_frame": flop48,
                                 "achieved": flop48 * fr48 / t_m / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": flop48 * fr48 / t_m / 1e12 / FP32_PEAK_TFLOPS}}
                # round 6: the analysis with a gradient back to the spectrogram (one autograd node: dsa_mcep_newton_update_bwd +
                # dsa_mcep_newton_resid_h_bwd per step; wall clock around whole calls, the backward is a sequence of launches)
                if fl != 1024:
                    with torch.enable_grad():
                        def fb48():
                            Xg = X48.detach().requires_grad_(True)
                            mc48(Xg).sum().backward()
                        rows48[list(rows48)[-1]]["mcep_fwd_bwd_ms"] = gpu_time(fb48, n=3, groups=2)
                        def sb48():
                            xg = x48.detach().requires_grad_(True)
                            torch.autograd.grad(st48(xg), xg, X48)
                        rows48[list(rows48)[-1]]["stft_fwd_bwd_ms"] = gpu_time(sb48, n=3, groups=2)   # (backward: csrc/stft_bwd_pk_big.h)
                        rows48[list(rows48)[-1]]["mcep_fwd_bwd_path"] = ("one autograd node (ops.McepNewtonStepsHFn)"
                                                                         if ops.mcep_newton_steps_grad_applies(m1_, mc48.D, mc48.E, mc48.alpha_vector)
                                                                         else "composed differentiable pieces")
                del X48
            del x48
    res["untuned_48khz"] = {"workload": "STFT (csrc/stft_pk_big.h since round 6) + MelCepstralAnalysis at the 48 kHz set-ups, 64 and 512 utterances x 1 s, "
                                        "float32, forward, n_iter=10; one workgroup per CU at 64 utterances (docs/DESIGN_LOG.md 3.41)", "rows": rows48,
                            "timing": "back-to-back calls (gpu_time)"}
    # ---- BASELINE configs[4] AS WRITTEN on one GPU: all 8 192 utterances through the one-launch step (what `--global-batch 8192`
    # times as the step at N = 1) ----
    try:
        with torch.no_grad():
            x8k = torch.randn(8192, SAMPLES, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
            fused = dsp.fuse(stft, mcep)
            fused(x8k)
            k8 = _lib.last_kernel()
            t8 = gpu_time(lambda: fused(x8k), n=5) * 1e-3
            t8_two = gpu_time(lambda: mcep(stft(x8k)), n=3) * 1e-3
        fr8 = 8192 * FRAMES_PER_UTT
        res["config5_whole_batch_one_gpu"] = {
            "workload": "BASELINE configs[4] as written at N = 1: 8 192 utterances x 1 s in ONE call of fuse(stft, mcep) (1 638 400 frames)",
            "kernel": k8, "path": fused.last_path, "ms_per_step": t8 * 1e3, "frames/s": fr8 / t8,
            "module_api_ms_per_step": t8_two * 1e3, "module_api_frames/s": fr8 / t8_two, "timing": "back-to-back calls (gpu_time)"}
        del x8k
    except Exception as e:
        res["config5_whole_batch_one_gpu"] = {"error": repr(e)}
    return res


DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 4096
_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")
_ROOF_OPTIONAL = ("algorithmic_frac", "back_to_back_ms", "frac_back_to_back", "step_period_ms", "frac_of_step_period")   # carried when the record has them


def speech_like_batch(B, dev):
    """"also run a speech-like input (data.wav tiled) because conditioning and Newton convergence differ".  The
    reference's test recording (tests/golden/datawav.npz: 19 200 int16 samples, read as diffsptk.read does: / 32768) repeated
    end to end; utterance u is the 16 000-sample stretch that starts 997 u samples into the repetition, so the 1 024 utterances
    are 1 024 different alignments of real speech (frames of silence, onsets and voiced stretches at every position).
    tests/test_gpu_configs.py checks the same batch against the C oracle at the bench size."""
    import numpy as np

    pcm = np.load(os.path.join(ROOT, "tests", "golden", "datawav.npz"))["pcm"].astype(np.float64) / 32768.0
    idx = (997 * np.arange(B)[:, None] + np.arange(SAMPLES)[None, :]) % pcm.size
    return torch.from_numpy(pcm[idx].astype(np.float32)).to(dev)


def _round(v, digits=5):
    """floats to `digits` significant digits (the line is a record, not a dump), containers recursively"""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    if isinstance(v, dict):
        return {k: _round(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_round(x, digits) for x in v]
    return v


def compact_line(res: dict, detail_path: str = DETAIL_FILE) -> str:
    """The ONE line rank 0 prints: first key "metric", the contract's scalar keys, `config`, `roofline` (dominant kernel),
    `roofline_stft` (the stage the north star puts a number on), `cpu_baseline` -- each reduced to the contract's own keys --
    and the path of the side file that holds everything else (sub-benchmarks of the other configs, notes, thread sweeps,
    counter records).  Guaranteed < LINE_LIMIT bytes: the driver keeps a bounded tail of stdout."""
    line = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in res}
    cfg = res.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "utterances_per_gpu", "global_batch", "frames_per_step", "parallelism",
                                          "rccl_world_size", "collective_backend", "reserved_cus", "gather_depth", "kernels", "path", "launches_per_step", "streams") if k in cfg}
    for name in ("roofline", "roofline_stft", "roofline_mcep"):
        r = res.get(name)
        if r:
            line[name] = {k: r.get(k) for k in _ROOF_KEYS}
            line[name].update({k: r[k] for k in _ROOF_OPTIONAL if r.get(k) is not None})
            if r.get("measured_in"):
                line[name]["measured_in"] = r["measured_in"]
    cb = res.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "threads_best", "kind", "sample")}
        if "gpu_over_cpu" in res:
            line["gpu_over_cpu"] = res["gpu_over_cpu"]
    for name in ("two_streams", "single_stream", "module_api", "speech_like"):   # the drop-in call mcep(stft(x)) beside the fused step; the speech-like input
        if res.get(name):
            line[name] = {k: v for k, v in res[name].items() if k != "note"}
    line["detail"] = detail_path
    s = json.dumps(_round(line), separators=(",", ":"))
    if len(s) >= LINE_LIMIT:   # never happens with the texts above; keep the contract anyway
        line["config"] = {"workload": str(cfg.get("workload", ""))[:300]}
        line.pop("speech_like", None)
        if "cpu_baseline" in line:
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample", ""))[:200]
        s = json.dumps(_round(line), separators=(",", ":"))
    assert len(s) < LINE_LIMIT and s.startswith('{"metric"'), len(s)
    return s


def emit(res: dict) -> str:
    """write the full record to bench_detail.json (repo root, and gpurun_out/ so that it comes back from a GPU box) and return
    the compact line"""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, DETAIL_FILE), "w") as f:
                json.dump(res, f, indent=1)
        except OSError:
            pass
    return compact_line(res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed steps run for this long before the warmup steps so that the GPU clocks have settled "
                         "(the first ~20 ms after idle run 10 %% slower; 0 disables)")
    ap.add_argument("--batch", type=int, default=1024, help="utterances per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: the whole job's utterances, split evenly over the N ranks (BASELINE configs[4] as written is "
                         "--global-batch 8192: N = 1 runs all 8192 utterances on one GPU, N = 8 runs 1024 each); 0 = weak scaling with --batch per GPU")
    ap.add_argument("--path", choices=["two-kernel", "fused"], default="fused",
                    help="the step: the ONE-launch STFT -> mel-cepstrum kernel (diffsptk_amd.fuse; 420 B/frame of memory traffic instead of "
                         "2476; the default since it measures faster: profiles/r04_fused_vs_two_kernels_v2.txt) or the fused STFT kernel "
                         "followed by the mel-cepstral kernel")
    ap.add_argument("--chunks", type=int, default=1,
                    help="N > 1: utterance chunks per step.  1 (default): one in-place all-gather per step, deferred "
                         "behind the next step's kernels; > 1: chunked gather/compute overlap inside the step")
    ap.add_argument("--streams", type=int, default=1,
                    help="consecutive steps (independent batches) alternate between this many streams.  1 (default): every launch runs "
                         "alone on the chip -- the timed region of rounds 1-5, and the one whose per-launch HIP events are a kernel's own "
                         "duration (what the roofline objects divide by).  2: every launch carries DSA_ALGO_OVERLAPPED_LAUNCHES -- its "
                         "short last round of tiles (204 800 frames = 6.25 rounds of the 2 048 wave slots) is packed onto 64 workgroups and "
                         "the other 192 CUs go to the next step's launch, which waits on the other stream (round 6: 0.5945 -> 0.5685 ms per "
                         "step at 200 steps, -2 % at 20).  At N = 1 the default run measures this mode too, AFTER the timed region, and "
                         "carries it in the line as `two_streams`")
    ap.add_argument("--record-every", type=int, default=4,
                    help="HIP events bracket the two launches of every n-th step of the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 2/3/4 sub-benchmarks (N = 1)")
    ap.add_argument("--algo", choices=["auto", "generic", "tuned"], default="auto")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
        # the loopback address) and pass rank 0's JSON line through; exactly what the documented torch.distributed.run
        # command line does
        import socket
        import subprocess

        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if os.environ.get("DSA_BENCH_SINGLE_DEVICE"):   # functional test of the N > 1 control flow on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DSA_BENCH_BACKEND", "nccl")   # "gloo" only for the one-GPU functional test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import diffsptk_amd as dsp
    from diffsptk_amd import _lib, ops
    from diffsptk_amd.dist import GATHER_DEPTH, analyze_chunked_overlap, reserved_cus

    algo = {"auto": _lib.ALGO_AUTO, "generic": _lib.ALGO_GENERIC, "tuned": _lib.ALGO_TUNED}[args.algo]
    B = args.batch
    strong = args.global_batch > 0
    if strong:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} ranks")
        B = args.global_batch // world
    x = torch.randn(B, SAMPLES, device=dev, generator=torch.Generator(device=dev).manual_seed(1234 + rank))
    stft = dsp.STFT(FL, FP, NFFT, device=dev)
    mcep = dsp.MelCepstralAnalysis(fft_length=NFFT, cep_order=M, alpha=ALPHA, n_iter=N_ITER, device=dev)

    n_chunks = 1 if world == 1 else args.chunks
    kernels = {}
    ev_log = []  # (stft_start, stft_end/mcep_start, mcep_end) HIP events of every launch pair in the timed region

    def compute(xc, record=False):
        """the hot path on one chunk of utterances; optionally bracketed by HIP events"""
        if record:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
        if args.path == "fused":   # one launch: the spectrogram never exists
            if record:
                e[1] = e[0]        # (one event in front of the launch: every record is a marker packet between two launches of the timed region)
            mc = ops.StftMcepFn.apply(xc, stft.window, stft.twiddle, mcep.G, mcep.D, mcep.E, mcep.alpha_vector, FL, FP, NFFT, True,
                                      1e-9, M, N_ITER)
            if record:
                e[2].record()
                ev_log.append((e, xc.size(0)))
            else:
                kernels["stft"] = kernels["mcep"] = _lib.last_kernel()
            return mc
        X = ops.StftFn.apply(xc, stft.window, stft.twiddle, FL, FP, NFFT, True, False, "constant", 1e-9, None, 3, algo)
        if record:
            e[1].record()
        else:
            kernels["stft"] = _lib.last_kernel()
        mc = ops.McepFn.apply(X, mcep.G, mcep.D, mcep.E, mcep.alpha_vector, NFFT, M, N_ITER, algo)
        if record:
            e[2].record()
            ev_log.append((e, xc.size(0)))
        else:
            kernels["mcep"] = _lib.last_kernel()
        return mc

    in_flight = []   # N > 1: (features, handle) of the last GATHER_DEPTH steps, whose all-gathers overlap this step

    # Steps are independent batches.  With --streams 2 they alternate between two side streams and every launch carries
    # DSA_ALGO_OVERLAPPED_LAUNCHES (ops.overlapped_launches): a launch's short last round is packed onto a quarter of the CUs and
    # the rest of the chip starts the next step's launch, which waits on the other stream.  Nothing is fenced inside the timed
    # region; the HIP events of a recorded step sit on that step's own stream, so a launch's duration includes the time it
    # shares the chip with its neighbours (what a rocprofv3 kernel trace of the same command reports too).
    n_streams = max(1, args.streams)
    main_stream = torch.cuda.current_stream()
    side_streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else []
    for s_ in side_streams:
        s_.wait_stream(main_stream)   # x was produced on the main stream
    step_no = [0]

    def step_body(record):
        # N > 1: the features are all-gathered (RCCL, one in-place all_gather_into_tensor) and the collective is left in
        # flight and completed GATHER_DEPTH = 2 steps later (a streaming consumer reads batch k while batch k+2 is computed), so it
        # hides behind the next steps' kernels instead of ending every step
        if world == 1:
            return analyze_chunked_overlap(x, lambda xc: compute(xc, record), n_chunks)
        out, handle = analyze_chunked_overlap(x, lambda xc: compute(xc, record), n_chunks, defer=True)
        in_flight.append((out, handle))
        while len(in_flight) > GATHER_DEPTH:   # completed TWO steps later: a gather starts in its step's tail and runs into the next launch
            in_flight.pop(0)[1].wait()     # (completed ONE step later the analysis and the exchange run in series: dist.reserved_cus)
        return out

    def step(record=False):
        if not side_streams:
            return step_body(record)
        step_no[0] += 1
        with torch.cuda.stream(side_streams[step_no[0] % n_streams]), ops.overlapped_launches():
            return step_body(record)

    def drain():
        while in_flight:
            in_flight.pop(0)[1].wait()

    gc.collect()   # (here, in front of the ramp: a collection right before the timed region would idle the chip and cost the clock)
    with torch.no_grad():
        if args.ramp_seconds > 0:   # clock ramp: untimed, same workload; a fixed count when N > 1 (collectives must match)
            t_ramp = time.perf_counter()
            rounds = 0
            while (time.perf_counter() - t_ramp < args.ramp_seconds) if world == 1 else (rounds < 30):
                for _ in range(10):
                    step()
                drain()
                torch.cuda.synchronize()
                rounds += 1
        for _ in range(max(args.warmup, 1)):
            step()
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        gc.disable()   # (as timeit does: no generation-2 collection -- tens of milliseconds of host time -- between two launches of the timed region)
        t0 = time.perf_counter()
        for i in range(args.steps):
            # HIP events bracket the two launches of every fourth step of the timed region only (the per-kernel
            # averages below come from those launches; with --streams > 1 those steps run alone)
            out = step(record=(i % max(1, args.record_every) == 0))
        drain()   # every collective of the K steps completes inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        gc.enable()
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert out.shape == (B * world, FRAMES_PER_UTT, M1) and bool(torch.isfinite(out).all())

    frames_rank = B * FRAMES_PER_UTT
    # average duration and frame count of ONE launch (a launch covers one chunk)
    t_stft = statistics.mean(e[0].elapsed_time(e[1]) for e, _ in ev_log) * 1e-3
    t_mcep = statistics.mean(e[1].elapsed_time(e[2]) for e, _ in ev_log) * 1e-3
    frames_launch = statistics.mean(nb for _, nb in ev_log) * FRAMES_PER_UTT
    stft_measured_in = "the timed region (HIP events around single launches)"
    k_stft_ref = None
    if args.path == "fused" and rank == 0:
        # The timed step is ONE launch whose prologue is the frame + window + rFFT stage: there is no STFT launch to bracket.  The
        # stage's own record (`roofline_stft`) comes from reference steps of the two-kernel path run AFTER the timed region,
        # bracketed exactly as the two-kernel path brackets them inside it (event, STFT, event, mel-cepstral kernel, event).
        ref_log = []
        nref = int(frames_launch // FRAMES_PER_UTT)
        with torch.no_grad():
            for i in range(12):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                Xr = ops.StftFn.apply(x[:nref], stft.window, stft.twiddle, FL, FP, NFFT, True, False, "constant", 1e-9, None, 3, algo)
                e[1].record()
                k_stft_ref = _lib.last_kernel()
                ops.McepFn.apply(Xr, mcep.G, mcep.D, mcep.E, mcep.alpha_vector, NFFT, M, N_ITER, algo)
                e[2].record()
                if i >= 2:
                    ref_log.append(e)
            torch.cuda.synchronize()
        t_stft = statistics.mean(e[0].elapsed_time(e[1]) for e in ref_log) * 1e-3
        t_mcep_two = statistics.mean(e[1].elapsed_time(e[2]) for e in ref_log) * 1e-3
        stft_measured_in = ("reference steps of the two-kernel path after the timed region (event, STFT, event, mel-cepstral kernel, event): "
                            "the timed step is ONE launch, this stage is its prologue and writes no spectrogram")
    if rank == 0:
        # the same two kernels back to back (no dispatch gap between an event and the launch): what a rocprofv3 kernel
        # trace reports as the kernels' own durations
        xl = x[: int(frames_launch // FRAMES_PER_UTT)]
        with torch.no_grad():
            Xl = stft(xl)
            t_stft_b2b = gpu_time(lambda: ops.StftFn.apply(xl, stft.window, stft.twiddle, FL, FP, NFFT, True, False, "constant",
                                                           1e-9, None, 3, algo), n=40) * 1e-3
            t_mcep_b2b = gpu_time(lambda: ops.McepFn.apply(Xl, mcep.G, mcep.D, mcep.E, mcep.alpha_vector, NFFT, M, N_ITER, algo),
                                  n=10) * 1e-3
        k_stft = k_stft_ref or kernels["stft"]   # (one-launch step: the stage's stand-alone kernel from the reference steps)
        pm_m, pm_s = pmc_static(kernels["mcep"]), pmc_static(k_stft)
        ipf = pm_m["derived"]["valu_insts_per_frame"] if pm_m else None
        res = {
            "metric": "frames/sec STFT->mcep (fl=400 fp=80 nfft=512 M=24)",
            "value": frames_rank * world * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[4] as written (strong scaling): STFT->mcep forward, {B * world} utterances x 1 s @ 16 kHz "
                             f"split over {world} GPU(s) = {B} per GPU ({frames_rank} frames each), alpha={ALPHA} n_iter={N_ITER}; "
                             "features all-gathered over RCCL when N>1") if strong else
                            (f"BASELINE configs[4] per-GPU shard: STFT->mcep forward, {B} utterances x 1 s @ 16 kHz "
                             f"per GPU ({frames_rank} frames), alpha={ALPHA} n_iter={N_ITER}; N=8 is the full "
                             "8192-utterance batch; features all-gathered over RCCL when N>1"),
                "path": args.path, "launches_per_step": 1 if args.path == "fused" else 2,
                "utterances_per_gpu": B, "global_batch": B * world, "frames_per_step": frames_rank * world,
                "parallelism": f"dp{world}", "rccl_world_size": (dist.get_world_size() if world > 1 else 1),
                "reserved_cus": (reserved_cus() if world > 1 else 0), "gather_depth": (GATHER_DEPTH if world > 1 else 0),   # CUs the persistent launches leave to RCCL's kernel (dist.reserved_cus)
                "collective_backend": (dist.get_backend() if world > 1 else None), "kernels": kernels, "chunks_per_step": n_chunks, "streams": n_streams,
                "arith": "float32 in / out / accumulate; the matrix chains of the mel-cepstral kernel run as 3-term "
                         "binary16 MFMA splits (hi/lo, dropped lo*lo: ~22-bit products), the STFT in packed float32",
            },
            "roofline": (lambda dp: {
                "kernel": kernels["mcep"], "bound": "f32_datapath (vector ALU + float32 matrix instructions share it)",
                "achieved": dp["achieved"] if dp else None, "peak": F32_DATAPATH_PEAK_GCPS, "unit": "G datapath-cycles/s",
                "frac": dp["frac"] if dp else None,
                "traffic": pmc_traffic(kernels["mcep"], frames_launch), "avg_launch_ms": t_mcep * 1e3,
                "algorithmic_frac": max((MCEP_F32_FLOP_PER_FRAME + (STFT_F32_FLOP_PER_FRAME if args.path == "fused" else 0)) / (FP32_PEAK_TFLOPS * 1e12),
                                        MCEP_F16_FLOP_PER_FRAME / (F16_MFMA_PEAK_TFLOPS * 1e12)) * frames_launch / t_mcep,
                "back_to_back_ms": t_mcep_b2b * 1e3, "frames_per_launch": frames_launch,
                "datapath": dp,
                "nominal_valu_issue": {"achieved": (ipf * frames_launch / t_mcep / 1e9) if ipf else None, "peak": VALU_ISSUE_PEAK_GIPS,
                                       "unit": "G wave-instr/s", "frac": (ipf * frames_launch / t_mcep / 1e9 / VALU_ISSUE_PEAK_GIPS) if ipf else None,
                                       "valu_insts_per_frame": ipf,
                                       "note": "vector wave-instructions (static: rocprofv3 SQ_INSTS_VALU, profiles/) against the guide's 2 cycles per "
                                               "wave64 instruction; the multiply-add class measures 4 cycles on this chip "
                                               "(profiles/r03_issue_calibration.txt), so this figure cannot reach 1 for FMA-heavy code"},
                "algorithmic": {"achieved": MCEP_FLOP_PER_FRAME * frames_launch / t_mcep / 1e12, "peak": FP32_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": MCEP_FLOP_PER_FRAME * frames_launch / t_mcep / 1e12 / FP32_PEAK_TFLOPS,
                                "flop_per_frame": MCEP_FLOP_PER_FRAME,
                                "note": "flops of the composed-matrix algorithm (DESIGN.md 3.2), each counted once; NOT a unit utilisation: "
                                        "the two matrix chains execute as binary16 products on the separate matrix pipe"},
                "pmc": pm_m["derived"] if pm_m else None, "pmc_source": (pm_m["_source"] + " (static)") if pm_m else None,
                "arith": "f16x3 split chains (fp32 accumulate) on the matrix pipe; 25x25 elimination as v_mfma_f32_4x4x1 rank-1 updates "
                         "+ fp32 VALU",
                "note": "achieved = float32-datapath cycles the kernel's instruction mix occupies (mix of the Newton-step loop read off the "
                        "code object, tools/isa_mix.py; per-instruction cycles measured with rocprofv3 --pmc on tools/bench_issue.cpp, "
                        "profiles/r03_issue_calibration.txt: multiply-add class 4, two-operand class 2, transcendental 8, 4x4x1 product 8) "
                        "x frames / measured launch time; peak = 1024 SIMDs x 2.4 GHz.  Two waves per SIMD (256 registers)",
                "power": {"socket_W_under_this_kernel": [1371, 1379], "socket_W_under_stft": [1316, 1333], "cap_W": 1400,
                          "source": "profiles/r03_power_probe.txt (static: tools/power_probe.sh, rocm-smi while the kernel loops)",
                          "note": "both headline kernels run at the socket's power cap; the kernel's cycle counter averages "
                                  "1.99-2.09 GHz over a launch, so the nominal-clock peak above is not reachable"},
            })(datapath_roofline(isa_mix("mcep_mfma_fwd_kernel_hILi8ELb1ELb0ELb0E" if args.path == "fused" else "mcep_mfma_fwd_kernel_hILi8ELb0ELb0ELb0E"), N_ITER, frames_launch, t_mcep)),
            "roofline_stft": {
                "kernel": k_stft, "bound": "hbm",
                "achieved": STFT_BYTES_PER_FRAME * frames_launch / t_stft / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": STFT_BYTES_PER_FRAME * frames_launch / t_stft / 1e9 / HBM_PEAK_GBS,
                "traffic": pmc_traffic(k_stft, frames_launch), "avg_launch_ms": t_stft * 1e3,
                "back_to_back_ms": t_stft_b2b * 1e3,
                "frac_back_to_back": STFT_BYTES_PER_FRAME * frames_launch / t_stft_b2b / 1e9 / HBM_PEAK_GBS,
                "pmc": pm_s["derived"] if pm_s else None, "pmc_source": (pm_s["_source"] + " (static)") if pm_s else None,
                "bytes_per_frame": STFT_BYTES_PER_FRAME,
                "measured_in": stft_measured_in,
                "note": "avg_launch_ms: HIP events around single launches (what `frac` uses), between two mel-cepstral launches; "
                        "back_to_back_ms: the same launch repeated between two events (module API, no mel-cepstral kernel in between)",
            },
        }
        if world == 1 and args.path == "fused":
            try:   # the other path of the same step, back to back (detail file only)
                with torch.no_grad():
                    t_two = gpu_time(lambda: mcep(stft(xl)), n=40) * 1e-3
                    fused_mod = dsp.fuse(stft, mcep)
                    t_fu = gpu_time(lambda: fused_mod(xl), n=40) * 1e-3
                res["two_kernel_path"] = {
                    "ms_per_step_back_to_back": t_two * 1e3, "frames/s": frames_launch / t_two, "fused_ms_back_to_back": t_fu * 1e3,
                    "mcep_kernel_ms_in_reference_steps": t_mcep_two * 1e3, "stft_kernel_ms_in_reference_steps": t_stft * 1e3,
                    "note": "bench.py --path two-kernel times this path as the step: `stft512_fwd` writes the (B, N, 257) power spectrogram, "
                            "`mcep_mfma_fwd` reads it back (1348 + 1128 algorithmic bytes per frame instead of 420)"}
            except Exception as e:
                res["two_kernel_path"] = {"error": repr(e)}
        if world == 1 and n_streams == 1 and args.path == "fused":
            # the same K steps alternating between TWO streams with DSA_ALGO_OVERLAPPED_LAUNCHES (bench.py --streams 2 times this as the
            # step): consecutive launches overlap at their ends -- more frames per second, while a launch's own [start, end] no longer
            # is its duration on the chip, which is why the timed region above (and `value`) stay single-stream
            try:
                s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
                for s_ in s2:
                    s_.wait_stream(main_stream)
                with torch.no_grad():
                    def ov(i):
                        with torch.cuda.stream(s2[i % 2]), ops.overlapped_launches():
                            return compute(x)
                    for i in range(40):   # (the two new streams' first launches carry one-off costs -- per-stream scratch, queue set-up:
                        ov(i)             #  with 6 warm-up steps a 20-step region measured 0.65 ms per step, 0.58 with the region run first)
                    torch.cuda.synchronize()
                    gc.disable()
                    t2 = time.perf_counter()
                    for i in range(args.steps):
                        ov(i)
                    torch.cuda.synchronize()
                    el2 = time.perf_counter() - t2
                    gc.enable()
                res["two_streams"] = {"ms_per_step": el2 / args.steps * 1e3, "value": frames_rank * args.steps / el2, "unit": "frames/s",
                                      "steps": args.steps,
                                      "note": "--streams 2: consecutive steps on alternating streams, DSA_ALGO_OVERLAPPED_LAUNCHES (a launch's short "
                                              "last round on 64 workgroups, the other CUs to the next launch); measured after the timed region"}
            except Exception as e:
                res["two_streams"] = {"error": repr(e)}
        if world == 1 and n_streams > 1:
            # the same K steps WITHOUT the overlap: one stream, every launch alone on the chip (the timed region of rounds 1-5)
            with torch.no_grad():
                for _ in range(5):
                    compute(x)
                torch.cuda.synchronize()
                gc.disable()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    compute(x)
                torch.cuda.synchronize()
                el1 = time.perf_counter() - t1
                gc.enable()
            res["single_stream"] = {"ms_per_step": el1 / args.steps * 1e3, "value": frames_rank * args.steps / el1, "unit": "frames/s",
                                    "steps": args.steps,
                                    "note": "--streams 1: each launch alone on the chip, its short last round (0.25 of 2 048 wave slots) one wave "
                                            "per SIMD; measured after the timed region, same process, same input"}
            r_ = res["roofline"]
            if r_.get("frac") is not None:
                # `frac` divides by the launch's own duration, which under the overlap includes the time it shares the chip with its
                # neighbours; the chip-level figure divides the same cycles by the step period
                r_["step_period_ms"] = res["ms_per_step"]
                r_["frac_of_step_period"] = r_["frac"] * r_["avg_launch_ms"] / res["ms_per_step"]
        if world == 1 and args.path == "fused" and "error" not in res.get("two_kernel_path", {}):
            # what a user of the reference's own call sequence gets: mcep(stft(x)) through the modules = two launches
            res["module_api"] = {"call": "mcep(stft(x))", "launches_per_step": 2, "ms_per_step": t_two * 1e3,
                                 "value": frames_launch / t_two, "unit": "frames/s", "fused_ms_back_to_back": t_fu * 1e3,
                                 "note": "back to back between two HIP events (bench.gpu_time), like fused_ms_back_to_back beside it"}
        if world == 1 and args.path == "fused":
            try:
                with torch.no_grad():
                    xs = speech_like_batch(B, dev)
                    fused_mod = dsp.fuse(stft, mcep)
                    mcs = fused_mod(xs)
                    t_sp = gpu_time(lambda: fused_mod(xs), n=40) * 1e-3
                    t_sp_two = gpu_time(lambda: mcep(stft(xs)), n=20) * 1e-3
                res["speech_like"] = {"workload": f"data.wav tiled: {B} x 1 s", "ms_per_step": t_sp * 1e3, "value": frames_rank / t_sp,
                                      "unit": "frames/s", "module_api_ms_per_step": t_sp_two * 1e3, "finite": bool(torch.isfinite(mcs).all()),
                                      "note": "same kernel, same launch; the arithmetic is data-independent (10 Newton steps whatever the "
                                              "conditioning), parity at this size: tests/test_gpu_configs.py::test_speech_like_batch_at_bench_size"}
                del xs, mcs
            except Exception as e:
                res["speech_like"] = {"error": repr(e)}
        if world == 1 and args.path != "fused":
            try:   # the other path of the same step, back to back (detail file only)
                fused_mod = dsp.fuse(stft, mcep)
                with torch.no_grad():
                    t_fu = gpu_time(lambda: fused_mod(xl), n=40) * 1e-3
                    k_fu = _lib.last_kernel()
                    t_two = gpu_time(lambda: mcep(stft(xl)), n=40) * 1e-3
                fb = FP * 4 + M1 * 4
                res["fused_path"] = {
                    "kernel": k_fu, "path": fused_mod.last_path, "ms_per_launch": t_fu * 1e3, "frames/s": frames_launch / t_fu,
                    "two_kernel_ms_back_to_back": t_two * 1e3,
                    "roofline": {"kernel": k_fu, "bound": "hbm", "bytes_per_frame": fb, "achieved": fb * frames_launch / t_fu / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fb * frames_launch / t_fu / 1e9 / HBM_PEAK_GBS,
                                 "traffic": pmc_traffic("stft512_mcep_fused_fwd", frames_launch), "avg_launch_ms": t_fu * 1e3},
                    "note": "diffsptk_amd.fuse(stft, mcep): ONE launch, 320 + 100 algorithmic bytes per frame (SURVEY 8(d)) instead of 1348 + 1128; "
                            "the persistent mel-cepstral wave computes its tile's 16 spectra itself.  The kernel is bound by the float32 datapath, "
                            "not by memory, and its FFT runs on scalar vector instructions (packed ones next to the other wave's 4x4x1 matrix "
                            "products return stale results, profiles/r04_fused_pk_hazard_check_v1.txt); it is the default step (--path)"}
            except Exception as e:
                res["fused_path"] = {"error": repr(e)}
        if world == 1 and not args.no_configs:
            try:
                res["configs"] = other_configs(dsp, ops, _lib, dev, stft, mcep, x[:1024] if x.size(0) >= 1024 else x)
            except Exception as e:   # the headline line must survive a failing sub-benchmark
                res["configs"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            try:
                res["cpu_baseline_c_oracle"] = c_oracle_baseline()
            except Exception as e:  # the oracle is optional test infrastructure
                res["cpu_baseline_c_oracle"] = {"error": str(e)}
        line = emit(res)
        print(line, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
