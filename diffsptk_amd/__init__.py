"""diffsptk_amd -- MI355X (gfx950) device backend for the STFT -> mel-cepstrum / LPC analysis
path of sp-nitech/diffsptk, behind the reference's own module / functional API.

    import diffsptk_amd as diffsptk
    stft = diffsptk.STFT(frame_length=400, frame_period=80, fft_length=512, device="cuda")
    mcep = diffsptk.MelCepstralAnalysis(fft_length=512, cep_order=24, alpha=0.42, n_iter=10, device="cuda")
    mc = mcep(stft(x))          # x: (..., T) on the GPU

All computation runs in hand-written HIP kernels (diffsptk_amd/csrc) through the C-ABI declared in
include/diffsptk_amd.h -- no vendor BLAS / FFT library is linked or called.  The only torch-operator
compositions are the `learnable=` options (modules/_learnable.py, as SURVEY.md 8(b) allows) and scalar
glue around kernels.  There is no CPU fallback.

    mc = diffsptk.fuse(stft, mcep)(x)   # the same in ONE launch (no spectrogram in memory)
    a = diffsptk.fuse(diffsptk.Frame(400, 80), diffsptk.Window(400, device="cuda"), diffsptk.LPC(400, 24, device="cuda"))(x)
                                        # lpc(window(frame(x))): one launch forward, one launch backward
"""
from . import functional
from .graph import Graphed
from .modules import *  # noqa: F401,F403
from .modules import __all__ as _module_names
from .utils.public import get_alpha, read, write

__version__ = "0.1.0"
__all__ = [*_module_names, "functional", "get_alpha", "read", "write", "Graphed"]
