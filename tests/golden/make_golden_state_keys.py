#!/usr/bin/env python3
"""state_dict keys and shapes of the REFERENCE's modules on the path (sp-nitech/diffsptk v4.0.0), learnable and not.

Runs ONLY in the build container (imports /root/reference with the two stub modules of make_golden.py); writes
tests/golden/state_keys.json: a list of {module, args, kwargs, state: {key: shape}} -- data, no reference text.
"""
import json
import os
import sys
import types

for name in ("torchaudio", "soundfile"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, "/root/reference")
import diffsptk as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    ("STFT", [400, 80, 512], {}),
    ("STFT", [400, 80, 512], {"learnable": True}),
    ("STFT", [400, 80, 512], {"learnable": ["basis"]}),
    ("STFT", [400, 80, 512], {"learnable": ["window"]}),
    ("STFT", [1200, 240, 2048], {"learnable": True}),
    ("STFT", [400, 80, 512], {"learnable": True, "out_format": "complex"}),
    ("Frame", [400, 80], {}),
    ("Window", [400, 512], {}),
    ("Window", [400, 512], {"learnable": True}),
    ("RealValuedFastFourierTransform", [512], {}),
    ("RealValuedFastFourierTransform", [512], {"learnable": True}),
    ("RealValuedInverseFastFourierTransform", [512], {"learnable": True}),
    ("Spectrum", [512], {}),
    ("FrequencyTransform", [24, 30, 0.42], {}),
    ("MelCepstralAnalysis", [], {"fft_length": 512, "cep_order": 24, "alpha": 0.42, "n_iter": 10}),
    ("Autocorrelation", [400, 24], {}),
    ("LevinsonDurbin", [24], {}),
    ("LinearPredictiveCodingAnalysis", [400, 24], {}),
    ("MelFilterBankAnalysis", [], {"fft_length": 512, "n_channel": 40, "sample_rate": 16000}),
    ("MelFilterBankAnalysis", [], {"fft_length": 512, "n_channel": 40, "sample_rate": 16000, "learnable": True}),
    ("MFCC", [], {"fft_length": 512, "mfcc_order": 12, "n_channel": 40, "sample_rate": 16000}),
    ("MFCC", [], {"fft_length": 512, "mfcc_order": 12, "n_channel": 40, "sample_rate": 16000, "learnable": True}),
    ("ISTFT", [400, 80, 512], {}),
    ("ISTFT", [400, 80, 512], {"learnable": True}),
    ("ISTFT", [400, 80, 512], {"learnable": ["window"]}),
    ("ISTFT", [400, 80, 512], {"learnable": ["basis"]}),
    ("Unframe", [400, 80], {}),
    ("Unframe", [400, 80], {"learnable": True}),
    ("MelGeneralizedCepstralAnalysis", [], {"fft_length": 512, "cep_order": 24, "alpha": 0.42, "gamma": -0.5, "n_iter": 3}),
    ("CepstralAnalysis", [], {"fft_length": 512, "cep_order": 24, "n_iter": 2}),
    ("PseudoMGLSADigitalFilter", [24, 80], {"alpha": 0.42}),
    ("PseudoMGLSADigitalFilter", [24, 80], {"alpha": 0.42, "learnable": True}),
    ("PseudoMGLSADigitalFilter", [[24, 10], 80], {"alpha": 0.42, "phase": "mixed", "learnable": True}),
    ("PseudoMGLSADigitalFilter", [24, 80], {"alpha": 0.42, "mode": "single-stage", "ir_length": 400, "n_fft": 512}),
    ("PseudoMGLSADigitalFilter", [24, 80], {"alpha": 0.42, "mode": "freq-domain", "frame_length": 400, "fft_length": 512}),
    ("MelGeneralizedCepstrumToSpectrum", [24, 512], {"alpha": 0.42}),
    ("MelCepstrumToMLSADigitalFilterCoefficients", [24], {"alpha": 0.42}),
    ("GriffinLim", [400, 80, 512], {"n_iter": 2}),
    ("DiscreteCosineTransform", [40], {}),
]


def main():
    out = []
    for name, args, kwargs in CASES:
        a = [tuple(v) if isinstance(v, list) else v for v in args]
        m = getattr(ref, name)(*a, **kwargs)
        out.append({"module": name, "args": args, "kwargs": kwargs,
                    "state": {k: list(v.shape) for k, v in m.state_dict().items()}})
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote {len(out)} cases")


if __name__ == "__main__":
    main()
