"""Reduction of the stale-result effect of DESIGN.md 3.25: one library variant (already copied over diffsptk_amd/lib/), the fused
STFT -> mel-cepstrum launch at the bench size with the spectrogram as a side product, compared bit for bit with the stand-alone
STFT kernel.  usage: python tools/hazard/hazard_check.py <variant-name> [reps] [n_iter]
Prints one line per launch: frames whose spectrogram differs (frames the launch never wrote -- ablation builds that idle half of the
waves -- are counted apart), the bin pattern of the first bad frames, and, for the self-check build (DSA_FUSED_DBG & 8), its log."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffsptk_amd as dsp
from diffsptk_amd import _lib, ops

name = sys.argv[1] if len(sys.argv) > 1 else "?"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda"
stft = dsp.STFT(400, 80, 512, device=dev)
B, T, N = 1024, 16000, 200
F = B * N
x = torch.randn(B, T, device=dev, generator=torch.Generator(device=dev).manual_seed(B))
scratch = torch.zeros(_lib.SCRATCH_BYTES, dtype=torch.uint8, device=dev)
mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=n_iter, device=dev)
images = ops.mcep_images(mcep.G, mcep.D, mcep.E, 512, 24)
selfcheck = "chk" in name
with torch.no_grad():
    X2 = stft(x).view(F, 257)
    for rep in range(reps):
        mc = torch.empty(F, 25, device=dev)
        X = torch.full((F, 257), float("nan"), device=dev)
        log = torch.zeros(16 + 16 * 4000, dtype=torch.int32, device=dev) if selfcheck else None
        ops._call("dsa_stft_mcep_fwd", x.data_ptr(), B, T, 400, 80, 512, stft.window.data_ptr(), stft.twiddle.data_ptr(), 1, 1e-9, 24, n_iter,
                  mcep.G.data_ptr(), mcep.D.data_ptr(), mcep.E.data_ptr(), mcep.alpha_vector.data_ptr(), _lib.F32, _lib.ALGO_AUTO,
                  images.data_ptr(), scratch.data_ptr(), mc.data_ptr(), None if log is None else log.data_ptr(), X.data_ptr(), ops._stream())
        torch.cuda.synchronize()
        untouched = torch.isnan(X).all(-1)
        diff = (X != X2) & ~untouched[:, None]
        bad = diff.any(-1).nonzero().flatten().cpu()
        print(f"[{name}] n_iter={n_iter} rep={rep}: bad frames {bad.numel()}  untouched {int(untouched.sum())}  "
              f"first tiles {sorted(set((bad // 16).tolist()))[:6]}", flush=True)
        for f in bad[:4].tolist():
            bins = diff[f].nonzero().flatten().cpu().tolist()
            mods = {m: sorted(set(b % m for b in bins)) for m in (2, 4, 16)}
            print(f"    frame {f} (tile {f // 16}, slot {f % 16}): {len(bins)} bins; mod2 {mods[2]} mod4 {mods[4]} mod16 {mods[16][:16]}; first {bins[:10]}")
        if log is not None:
            lg = log.cpu()
            cnt = int(lg[0])
            print(f"    self-check log: {cnt} mismatching lane-transforms")
            import struct
            def fl(u):
                return struct.unpack("<f", struct.pack("<I", int(u) & 0xffffffff))[0]
            simds = {}
            for s in range(min(cnt, 4000)):
                r = lg[16 + 16 * s: 32 + 16 * s].tolist()
                hw = r[11] & 0xffffffff
                key = (r[12] & 0xf, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3)
                simds[key] = simds.get(key, 0) + 1
                if s < 24:
                    print(f"      tile {r[0]} pass {r[1]} lane {r[2]} reg {r[3]} nbad {r[4]}: pk ({fl(r[5]):.6g},{fl(r[6]):.6g}) sc ({fl(r[7]):.6g},{fl(r[8]):.6g}) "
                          f"redo ({fl(r[9]):.6g},{fl(r[10]):.6g}) redo==sc {r[9] == r[7] and r[10] == r[8]} wave {r[13]} wg {r[14]} xcc {r[12] & 0xf} hwid {hw:08x}")
            print(f"    (xcc, se, cu, simd) -> count: {dict(sorted(simds.items(), key=lambda kv: -kv[1])[:12])}  distinct {len(simds)}")
