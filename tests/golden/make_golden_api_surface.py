#!/usr/bin/env python3
"""The reference's API surface for the path: constructor / forward signatures of the exported classes, signatures of the functionals,
and what 69 invalid option sets raise (exception type + text).  Build container only; writes tests/golden/api_surface.json (data)."""
import inspect
import json
import os
import sys
import types

for name in ("torchaudio", "soundfile"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.path.insert(0, "/root/reference")
import diffsptk as ref  # noqa: E402
import diffsptk.functional as RF  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
# the names this repo exports (its __all__), looked up in the reference
NAMES = json.load(open(os.path.join(HERE, "api_names.json")))


def sig(f):
    return [[p.name, p.kind.name, None if p.default is inspect._empty else repr(p.default)] for p in inspect.signature(f).parameters.values()]


BAD = [
    ("Frame", [0, 80], {}), ("Frame", [400, 0], {}),
    ("Window", [0], {}), ("Window", [400, 300], {}), ("Window", [400], {"window": "foo"}), ("Window", [400], {"norm": "foo"}), ("Window", [400], {"window": 9}), ("Window", [400], {"norm": 5}),
    ("RealValuedFastFourierTransform", [0], {}), ("RealValuedFastFourierTransform", [511], {}), ("RealValuedFastFourierTransform", [512], {"out_format": "foo"}),
    ("Spectrum", [0], {}), ("Spectrum", [512], {"eps": -1}), ("Spectrum", [512], {"relative_floor": 1}), ("Spectrum", [512], {"out_format": "foo"}), ("Spectrum", [511], {}),
    ("STFT", [400, 80, 256], {}), ("STFT", [400, 80, 512], {"learnable": "x"}), ("STFT", [400, 80, 512], {"learnable": ["foo"]}), ("STFT", [400, 80, 512], {"out_format": "foo"}),
    ("FrequencyTransform", [-1, 3, 0.1], {}), ("FrequencyTransform", [3, -1, 0.1], {}), ("FrequencyTransform", [3, 3, 1.0], {}),
    ("MelCepstralAnalysis", [], {"fft_length": 512, "cep_order": -1, "alpha": 0.42}), ("MelCepstralAnalysis", [], {"fft_length": 512, "cep_order": 300, "alpha": 0.42}),
    ("MelCepstralAnalysis", [], {"fft_length": 512, "cep_order": 24, "alpha": 1.0}), ("MelCepstralAnalysis", [], {"fft_length": 512, "cep_order": 24, "alpha": 0.42, "n_iter": -1}),
    ("Autocorrelation", [0, 24], {}), ("Autocorrelation", [400, 400], {}), ("Autocorrelation", [400, 24], {"out_format": "foo"}), ("Autocorrelation", [400, -1], {}),
    ("LevinsonDurbin", [-1], {}), ("LevinsonDurbin", [24], {"eps": -1.0}),
    ("LinearPredictiveCodingAnalysis", [400, 400], {}), ("LinearPredictiveCodingAnalysis", [0, 24], {}),
    ("MelFilterBankAnalysis", [], {"fft_length": 512, "n_channel": 0, "sample_rate": 16000}), ("MelFilterBankAnalysis", [], {"fft_length": 512, "n_channel": 40, "sample_rate": 0}),
    ("MelFilterBankAnalysis", [], {"fft_length": 512, "n_channel": 40, "sample_rate": 16000, "f_min": 9000}), ("MelFilterBankAnalysis", [], {"fft_length": 512, "n_channel": 40, "sample_rate": 16000, "floor": 0}),
    ("MelFilterBankAnalysis", [], {"fft_length": 512, "n_channel": 40, "sample_rate": 16000, "scale": "foo"}), ("MelFilterBankAnalysis", [], {"fft_length": 512, "n_channel": 40, "sample_rate": 16000, "out_format": "foo"}),
    ("MFCC", [], {"fft_length": 512, "mfcc_order": 41, "n_channel": 40, "sample_rate": 16000}), ("MFCC", [], {"fft_length": 512, "mfcc_order": 0, "n_channel": 40, "sample_rate": 16000}),
    ("MFCC", [], {"fft_length": 512, "mfcc_order": 12, "n_channel": 40, "sample_rate": 16000, "lifter": 0}),
    ("ISTFT", [400, 80, 256], {}), ("ISTFT", [400, 80, 512], {"learnable": ["foo"]}),
    ("Unframe", [0, 80], {}),
    ("MelGeneralizedCepstralAnalysis", [], {"fft_length": 512, "cep_order": 24, "alpha": 0.42, "gamma": 0.5}), ("MelGeneralizedCepstralAnalysis", [], {"fft_length": 512, "cep_order": 300}),
    ("CepstralAnalysis", [], {"fft_length": 512, "cep_order": 300}), ("CepstralAnalysis", [], {"fft_length": 512, "cep_order": 24, "accel": -1}), ("CepstralAnalysis", [], {"fft_length": 512, "cep_order": 24, "n_iter": -1}),
    ("PseudoMGLSADigitalFilter", [24, 80], {"phase": "foo"}), ("PseudoMGLSADigitalFilter", [24, 80], {"mode": "foo"}), ("PseudoMGLSADigitalFilter", [24, 80], {"taylor_order": -1}),
    ("MelGeneralizedCepstrumToSpectrum", [24, 512], {"out_format": "foo"}), ("MelGeneralizedCepstrumToSpectrum", [-1, 512], {}), ("MelGeneralizedCepstrumToSpectrum", [24, 1], {}),
    ("MelCepstrumToMLSADigitalFilterCoefficients", [-1], {}), ("MelCepstrumToMLSADigitalFilterCoefficients", [24], {"alpha": 1.0}),
    ("GriffinLim", [400, 80, 512], {"n_iter": -1}), ("GriffinLim", [400, 80, 512], {"alpha": -1}), ("GriffinLim", [400, 80, 512], {"init_phase": "foo"}),
    ("DiscreteCosineTransform", [0], {}), ("DiscreteCosineTransform", [40], {"dct_type": 5}),
]


def main():
    out = {"classes": {}, "functions": {}, "functional": {}, "errors": []}
    for n in NAMES["package"]:
        o = getattr(ref, n, None)
        if o is None:
            continue
        if inspect.isclass(o):
            out["classes"][n] = {"init": sig(o.__init__), "forward": sig(o.forward)}
        elif callable(o):
            out["functions"][n] = sig(o)
    for n in NAMES["functional"]:
        o = getattr(RF, n, None)
        if o is not None and inspect.isfunction(o):
            out["functional"][n] = sig(o)
    for name, args, kw in BAD:
        try:
            getattr(ref, name)(*args, **kw)
            res = ["ok", ""]
        except Exception as e:   # noqa: BLE001
            res = [type(e).__name__, str(e)]
        out["errors"].append({"module": name, "args": args, "kwargs": kw, "raises": res})
    json.dump(out, open(os.path.join(HERE, "api_surface.json"), "w"), separators=(",", ":"))
    print(len(out["classes"]), "classes,", len(out["functional"]), "functionals,", len(out["errors"]), "error cases")


if __name__ == "__main__":
    main()
