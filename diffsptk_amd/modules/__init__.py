from .acorr import Autocorrelation
from .base import BaseFunctionalModule, Precomputed
from .fftr import RealValuedFastFourierTransform
from .frame import Frame
from .freqt import FrequencyTransform
from .levdur import LevinsonDurbin
from .lpc import LinearPredictiveCodingAnalysis
from .lpc import LinearPredictiveCodingAnalysis as LPC
from .mcep import MelCepstralAnalysis
from .spec import Spectrum
from .stft import ShortTimeFourierTransform
from .stft import ShortTimeFourierTransform as STFT
from .window import Window

__all__ = [
    "Autocorrelation", "BaseFunctionalModule", "Frame", "FrequencyTransform", "LPC", "LevinsonDurbin",
    "LinearPredictiveCodingAnalysis", "MelCepstralAnalysis", "Precomputed",
    "RealValuedFastFourierTransform", "STFT", "ShortTimeFourierTransform", "Spectrum", "Window",
]
