from .acorr import Autocorrelation
from .base import BaseFunctionalModule, Precomputed
from .dct import DiscreteCosineTransform
from .dct import DiscreteCosineTransform as DCT
from .fbank import MelFilterBankAnalysis
from .fbank import MelFilterBankAnalysis as FBANK
from .fftcep import CepstralAnalysis
from .fftr import RealValuedFastFourierTransform
from .frame import Frame
from .griffin import GriffinLim
from .ifftr import RealValuedInverseFastFourierTransform
from .istft import InverseShortTimeFourierTransform
from .istft import InverseShortTimeFourierTransform as ISTFT
from .freqt import FrequencyTransform
from .levdur import LevinsonDurbin
from .lpc import LinearPredictiveCodingAnalysis
from .lpc import LinearPredictiveCodingAnalysis as LPC
from .fused import FusedFrameWindowLPC, FusedSTFTFilterBank, FusedSTFTMelCepstralAnalysis, fuse
from .gnorm import GeneralizedCepstrumGainNormalization, GeneralizedCepstrumInverseGainNormalization
from .mc2b import MelCepstrumToMLSADigitalFilterCoefficients, MLSADigitalFilterCoefficientsToMelCepstrum
from .mcep import MelCepstralAnalysis
from .mgc2mgc import MelGeneralizedCepstrumToMelGeneralizedCepstrum
from .mgc2sp import MelGeneralizedCepstrumToSpectrum
from .mgcep import MelGeneralizedCepstralAnalysis
from .mglsadf import PseudoMGLSADigitalFilter
from .mglsadf import PseudoMGLSADigitalFilter as MLSA
from .zerodf import AllZeroDigitalFilter, LinearInterpolation
from .mfcc import MelFrequencyCepstralCoefficientsAnalysis
from .mfcc import MelFrequencyCepstralCoefficientsAnalysis as MFCC
from .spec import Spectrum
from .stft import ShortTimeFourierTransform
from .stft import ShortTimeFourierTransform as STFT
from .unframe import Unframe
from .window import Window

__all__ = [
    "Autocorrelation", "BaseFunctionalModule", "CepstralAnalysis", "DCT", "DiscreteCosineTransform", "FBANK", "Frame", "GriffinLim",
    "FrequencyTransform", "ISTFT", "InverseShortTimeFourierTransform", "RealValuedInverseFastFourierTransform", "Unframe", "LPC", "LevinsonDurbin", "LinearPredictiveCodingAnalysis", "MFCC", "MelCepstralAnalysis",
    "MelFilterBankAnalysis", "MelFrequencyCepstralCoefficientsAnalysis", "Precomputed",
    "GeneralizedCepstrumGainNormalization", "GeneralizedCepstrumInverseGainNormalization",
    "MelCepstrumToMLSADigitalFilterCoefficients", "MLSADigitalFilterCoefficientsToMelCepstrum",
    "MelGeneralizedCepstrumToMelGeneralizedCepstrum", "MelGeneralizedCepstrumToSpectrum", "MelGeneralizedCepstralAnalysis",
    "PseudoMGLSADigitalFilter", "MLSA", "AllZeroDigitalFilter", "LinearInterpolation",
    "FusedFrameWindowLPC", "FusedSTFTFilterBank", "FusedSTFTMelCepstralAnalysis", "fuse",
    "RealValuedFastFourierTransform", "STFT", "ShortTimeFourierTransform", "Spectrum", "Window",
]
