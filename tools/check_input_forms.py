"""Input FORMS the reference's ATen ops accept: non-contiguous views, any number of leading dims (none included), float64, inputs
longer / shorter than fft_length for fftr (torch.fft.rfft(x, n) truncates / zero-pads).  Layout cases must equal the contiguous
copy's result bit for bit; fftr lengths are compared with torch.fft.rfft on the host.

    python tools/check_input_forms.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsptk_amd as dsp  # noqa: E402

dev = torch.device("cuda", 0)


def run_all():
    bad = 0

    def report(tag, ok, extra=""):
        nonlocal bad
        bad += not ok
        print(f"  {tag}: {'ok' if ok else 'MISMATCH'} {extra}", flush=True)

    def guard(tag, fn):
        nonlocal bad
        try:
            fn()
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"  {tag}: FAILED with {type(e).__name__}: {e}", flush=True)

    g = torch.Generator(device=dev).manual_seed(3)
    stft = dsp.STFT(400, 80, 512, device=dev)
    mcep = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev)
    fused = dsp.fuse(stft, mcep)
    frame, window, lpc = dsp.Frame(400, 80), dsp.Window(400, device=dev), dsp.LPC(400, 24, device=dev)
    flpc = dsp.fuse(frame, window, lpc)
    print("layouts:", flush=True)
    base = torch.randn(6, 8000, device=dev, generator=g)
    views = {
        "every second sample (stride 2)": torch.randn(6, 16000, device=dev, generator=g)[:, ::2],
        "transposed storage": torch.randn(8000, 6, device=dev, generator=g).t(),
        "expanded (stride 0 batch)": base[:1].expand(6, 8000),
        "offset slice": torch.randn(6, 9000, device=dev, generator=g)[:, 37:8037],
        "one-dimensional": base[0],
        "three leading dims": base.reshape(1, 2, 3, 8000),
    }
    for tag, v in views.items():
        def one(v=v, tag=tag):
            c = v.contiguous()
            for name, fn in (("stft", stft), ("mcep(stft)", lambda z: mcep(stft(z))), ("fuse(stft, mcep)", fused),
                             ("lpc(window(frame))", lambda z: lpc(window(frame(z)))), ("fuse(frame, window, lpc)", flpc)):
                a, b = fn(v), fn(c)
                report(f"{tag}: {name} shape {tuple(a.shape)}", a.shape == b.shape and torch.equal(a, b))
            # gradient w.r.t. a non-contiguous leaf-view
            vv = v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
            cc = c.detach().clone().requires_grad_(True)
            fused(vv).square().sum().backward()
            fused(cc).square().sum().backward()
            report(f"{tag}: gradient of fuse(stft, mcep)", torch.equal(vv.grad.contiguous(), cc.grad.reshape(vv.grad.shape)))
        guard(tag, one)
    print("non-contiguous spectrogram into mcep / cotangent into backward:", flush=True)

    def noncontig_spec():
        X = stft(base)                                     # (6, 100, 257)
        Xt = X.transpose(0, 1).contiguous().transpose(0, 1)   # same values, other strides
        report("mcep on a permuted-stride spectrogram", torch.equal(mcep(Xt), mcep(X)))
        xg = base.clone().requires_grad_(True)
        y = mcep(stft(xg))
        gy = torch.randn(100, 6, 25, device=dev, generator=g).transpose(0, 1)   # non-contiguous cotangent
        y.backward(gy)
        xg2 = base.clone().requires_grad_(True)
        mcep(stft(xg2)).backward(gy.contiguous())
        report("backward with a non-contiguous cotangent", torch.equal(xg.grad, xg2.grad))
    guard("non-contiguous spectrogram", noncontig_spec)
    print("float64 through the path (generic kernels):", flush=True)

    def f64():
        s64 = dsp.STFT(400, 80, 512, device=dev, dtype=torch.float64)
        m64 = dsp.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device=dev, dtype=torch.float64)
        x64 = base.double()
        y64 = m64(s64(x64))
        y32 = mcep(stft(base))
        err = float((y64 - y32.double()).abs().max())
        report(f"mcep(stft) float64 {tuple(y64.shape)} {y64.dtype}", y64.dtype == torch.float64 and err < 5e-5, f"(max |f64 - f32| = {err:.2e})")
        try:
            m64(stft(base))
            report("float32 spectrogram into a float64 module raises", False)
        except (RuntimeError, TypeError) as e:
            report("float32 spectrogram into a float64 module raises", True, f"({type(e).__name__})")
    guard("float64", f64)
    print("fftr with inputs shorter / longer than fft_length (torch.fft.rfft(x, n) pads / truncates):", flush=True)
    for fmt in ("complex", "power"):
        fftr = dsp.RealValuedFastFourierTransform(512, out_format=fmt, device=dev)
        for Lx in (1, 300, 511, 512, 513, 700):
            def one(fmt=fmt, Lx=Lx, fftr=fftr):
                x = torch.randn(5, Lx, device=dev, generator=g)
                y = fftr(x)
                r = torch.fft.rfft(x.cpu().double(), n=512)
                r = r if fmt == "complex" else r.abs().square()
                yy = torch.view_as_complex(y.contiguous()) if (fmt == "complex" and not y.is_complex()) else y
                err = float((yy.cpu().to(r.dtype) - r).abs().max() / r.abs().max())
                report(f"fftr {fmt}, input length {Lx}", tuple(yy.shape) == tuple(r.shape) and err < 1e-5, f"(rel. error {err:.1e})")
            guard(f"fftr {fmt} length {Lx}", one)
    print("mismatching / failing checks:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run_all() else 0)
