"""Loader / builder of the C-ABI shared library (include/diffsptk_amd.h).

The library is built IN-TREE by hipcc for gfx950 (``build()``; ``__graft_entry__.build()`` calls
it) and loaded with ctypes.  There is no CPU fallback: if the library is missing, or no HIP
device is visible, the ops raise -- they never silently compute elsewhere.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdiffsptk_amd.so")
SOURCES = ("stft.hip", "mcep.hip", "mcep_mfma.hip", "lpc.hip", "fbank.hip", "fftcep.hip", "mgc.hip", "rows_gemm.hip", "thsolve_quad.hip")
HIPCC_FLAGS = (
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-mcode-object-version=5", "-Wno-unused-value", "-ffp-contract=on",
)
# Per-source additions.  stft.hip: the compiler's automatic v_pk_*_f32 selection costs the register-FFT kernels
# more in register-pairing moves than it saves (forward 88 -> 82 us per 204 800 frames without it; the packed
# kernels of stft_pk.h / stft_bwd_pk.h switch the feature back on for themselves and place v_pk_* by hand).
# Round 6: the same for every unit whose compiler-made packed code contained forms with a set op_sel bit (a low result half
# reading a high source half: the instruction class of DESIGN.md 4, which no shipped kernel may execute --
# tests/test_host_cpu.py::test_no_crossed_packed_float32); kernels with hand-placed packed instructions carry DSA_PK_TARGET.
_NO_PK = ("-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops")
SOURCE_FLAGS = {
    "stft.hip": _NO_PK,
    "fbank.hip": _NO_PK,
    "mgc.hip": _NO_PK,
    "thsolve_quad.hip": _NO_PK,
    "mcep_mfma.hip": _NO_PK,   # (its kernels carry DSA_PK_TARGET -- measured faster with the compiler's pairing -- except mgcep_step_h)
}

F32, F64 = 0, 1
SCRATCH_BYTES = 64   # DSA_SCRATCH_BYTES
FBANK_PLAN_FLOATS = 2048   # DSA_FBANK_PLAN_FLOATS
ERR_UNSUPPORTED = -2       # DSA_ERR_UNSUPPORTED
LPC_EXACT_LAGSUMS = 0x200  # DSA_LPC_EXACT_LAGSUMS
ALGO_AUTO, ALGO_GENERIC, ALGO_TUNED = 0, 1, 2
ALGO_SCRATCH_IS_CLEAN = 0x100   # DSA_ALGO_SCRATCH_IS_CLEAN
ALGO_SCRATCH_HAS_WORKSPACE = 0x200   # DSA_ALGO_SCRATCH_HAS_WORKSPACE
ALGO_HIST_HAS_RT = 0x400             # DSA_ALGO_HIST_HAS_RT
ALGO_OVERLAPPED_LAUNCHES = 0x800     # DSA_ALGO_OVERLAPPED_LAUNCHES


def algo_reserve_cus(n: int) -> int:
    """DSA_ALGO_RESERVE_CUS(n)"""
    return (int(n) & 63) << 16
MCEP_BWD_WORKSPACE_BYTES = SCRATCH_BYTES + 512 * 16 * 32 * 4   # DSA_MCEP_BWD_WORKSPACE_BYTES

_lib = None


class BackendError(RuntimeError):
    """The HIP library is missing, failed to load, or a call returned an error status."""


def _sources():
    files = [os.path.join(CSRC, s) for s in SOURCES]
    deps = files + [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + [os.path.join(_ROOT, "include", "diffsptk_amd.h"),
                    os.path.abspath(__file__)]   # the build flags live in this file
    return files, deps


def is_stale() -> bool:
    _, deps = _sources()
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library (cross-compiles w/o GPU)."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise BackendError("hipcc not found: cannot build libdiffsptk_amd.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    files, _ = _sources()
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    objs, procs = [], []
    for f in files:   # one object per source (its own flags), compiled concurrently, then one link
        obj = os.path.join(obj_dir, os.path.basename(f) + ".o")
        cmd = [hipcc, *HIPCC_FLAGS, *SOURCE_FLAGS.get(os.path.basename(f), ()), "-c", f, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise BackendError("hipcc failed: " + " ".join(cmd) + "\n" + out)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", tmp, *objs]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise BackendError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    global _lib
    _lib = None
    return LIB_PATH


# name -> (restype, argtypes); mirrors include/diffsptk_amd.h
_P, _I, _L, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SIGNATURES = {
    "dsa_version": (C.c_int, []),
    "dsa_last_error": (C.c_char_p, []),
    "dsa_device_count": (C.c_int, []),
    "dsa_last_kernel": (C.c_char_p, []),
    "dsa_num_frames": (C.c_int64, [_L, _I]),
    "dsa_frame_fwd": (C.c_int, [_P, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dsa_frame_bwd": (C.c_int, [_P, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dsa_window_fwd": (C.c_int, [_P, _L, _I, _P, _I, _I, _P, _P]),
    "dsa_window_bwd": (C.c_int, [_P, _P, _L, _I, _P, _I, _I, _P, _P, _P]),
    "dsa_fftr_fwd": (C.c_int, [_P, _L, _I, _I, _I, _P, _I, _P, _P]),
    "dsa_fftr_bwd": (C.c_int, [_P, _P, _L, _I, _I, _I, _P, _I, _P, _P]),
    "dsa_spec_fwd": (C.c_int, [_P, _I, _P, _I, _L, _I, _D, _I, _D, _I, _P, _I, _P, _P]),
    "dsa_spec_bwd": (C.c_int, [_P, _P, _I, _P, _I, _L, _I, _D, _I, _D, _I, _P, _I, _P, _P, _P]),
    "dsa_stft_fwd": (C.c_int, [_P, _L, _L, _I, _I, _I, _P, _P, _I, _I, _I, _D, _I, _D, _I, _I, _I, _P, _P]),
    "dsa_stft_bwd": (C.c_int, [_P, _P, _L, _L, _I, _I, _I, _P, _P, _I, _I, _I, _D, _I, _D, _I, _I, _I, _P, _P, _P]),
    "dsa_freqt_fwd": (C.c_int, [_P, _L, _I, _P, _I, _I, _P, _P]),
    "dsa_freqt_bwd": (C.c_int, [_P, _L, _I, _P, _I, _I, _P, _P]),
    "dsa_rows_gemm": (C.c_int, [_P, _L, _I, _P, _I, _I, _I, _P, _I, _I, _P, _I, _P]),
    "dsa_mcep_newton_update": (C.c_int, [_P, _L, _I, _P, _I, _P, _P, _P]),
    "dsa_gnorm_fwd": (C.c_int, [_P, _L, _I, _D, _I, _I, _P, _P]),
    "dsa_mgcep_gain": (C.c_int, [_P, _P, _P, _L, _I, _D, _I, _P, _P]),
    "dsa_mcep_newton_update_bwd": (C.c_int, [_P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "dsa_mcep_newton_resid": (C.c_int, [_P, _L, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P]),
    "dsa_mcep_resid_images_bytes": (C.c_int64, [_I, _I]),
    "dsa_mcep_resid_prepare": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _P, _P]),
    "dsa_mcep_newton_resid_h": (C.c_int, [_P, _L, _I, _P, _I, _P, _I, _P, _P]),
    "dsa_mcep_resid_bwd_images_bytes": (C.c_int64, [_I, _I]),
    "dsa_mcep_resid_bwd_prepare": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _P, _P]),
    "dsa_mcep_newton_resid_h_bwd": (C.c_int, [_P, _L, _I, _P, _I, _P, _P, _I, _P, _P, _P]),
    "dsa_mcep_newton_glogx_h": (C.c_int, [_P, _L, _I, _P, _I, _P, _I, _P, _I, _P, _P]),
    "dsa_mcep_newton_steps": (C.c_int, [_P, _L, _I, _P, _I, _P, _P, _I, _I, _P, _P]),
    "dsa_rows_ew": (C.c_int, [_I, _I, _P, _P, _P, _L, _I, _P, _P, _P]),
    "dsa_irfft_scale": (C.c_int, [_P, _L, _I, _I, _P, _P]),
    "dsa_div_rows": (C.c_int, [_P, _L, _L, _P, _D, _I, _P, _P]),
    "dsa_istft_fwd": (C.c_int, [_P, _L, _L, _I, _I, _I, _P, _P, _I, _P, _D, _I, _I, _P, _P]),
    "dsa_fbank_dct_fwd": (C.c_int, [_P, _L, _I, _P, _I, _P, _I, _D, _D, _I, _I, _P, _P, _P]),
    "dsa_fftcep_fwd": (C.c_int, [_P, _L, _I, _I, _P, _D, _I, _I, _P, _P, _P]),
    "dsa_fftcep_bwd": (C.c_int, [_P, _P, _L, _I, _I, _P, _D, _I, _P, _I, _P, _P]),
    "dsa_griffin_update": (C.c_int, [_P, _L, _L, _L, _I, _P, _P, _P, _P, _I, _D, _D, _D, _D, _I, _P, _P]),
    "dsa_fbank_fwd": (C.c_int, [_P, _L, _I, _P, _I, _D, _D, _I, _I, _P, _P, _P]),
    "dsa_fbank_scan_plan": (C.c_int, [_P, _I, _I, _P]),
    "dsa_mgcep_spectra": (C.c_int, [_P, _P, _L, _I, _I, _P, _P, _D, _I, _P, _P]),
    "dsa_stft_fbank_fwd": (C.c_int, [_P, _L, _L, _I, _I, _I, _P, _P, _I, _D, _P, _I, _D, _D, _I, _I, _P, _P]),
    "dsa_fbank_bins_plan": (C.c_int, [_P, _I, _I, _P]),
    "dsa_fbank_bins_bwd": (C.c_int, [_P, _P, _L, _I, _I, _P, _D, _D, _I, _P, _P]),
    "dsa_fbank_bwd": (C.c_int, [_P, _P, _P, _L, _I, _P, _I, _D, _D, _I, _I, _P, _P]),
    "dsa_mcep_images_bytes": (C.c_int64, [_I, _I, _I]),
    "dsa_mcep_prepare": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "dsa_mcep_fwd": (C.c_int, [_P, _L, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "dsa_stft_mcep_fwd": (C.c_int, [_P, _L, _L, _I, _I, _I, _P, _P, _I, _D, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "dsa_stft_mcep_opts_fwd": (C.c_int, [_P, _L, _L, _I, _I, _I, _P, _P, _I, _I, _I, _D, _I, _D, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P,
                                         _P, _P]),
    "dsa_mcep_bwd": (C.c_int, [_P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "dsa_thsolve_fwd": (C.c_int, [_P, _P, _P, _L, _I, _I, _P, _P]),
    "dsa_mgcep_step": (C.c_int, [_P, _P, _L, _I, _I, _D, _P, _I, _P, _P, _P, _P]),
    "dsa_mgcep_step_solve": (C.c_int, [_P, _P, _L, _I, _I, _D, _P, _I, _P, _P, _P, _P, _I, _P, _P]),
    "dsa_mgcep_step_bwd_h": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _D, _P, _I, _P, _P, _P, _P]),
    "dsa_mgcep_step_bwd": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _D, _P, _I, _P, _P, _P, _P]),
    "dsa_gc2gc_fwd": (C.c_int, [_P, _L, _I, _I, _D, _D, _I, _P, _I, _I, _P, _P]),
    "dsa_gc2gc_bwd": (C.c_int, [_P, _P, _L, _I, _I, _D, _D, _I, _P, _I, _P, _P]),
    "dsa_thsolve_bwd": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "dsa_thsolve_update_fwd": (C.c_int, [_P, _P, _P, _L, _L, _L, _I, _I, _P, _P, _P]),
    "dsa_zerodf_fwd": (C.c_int, [_P, _P, _L, _L, _I, _I, _I, _I, _I, _P, _P]),
    "dsa_zerodf_bwd": (C.c_int, [_P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _P, _P, _P]),
    "dsa_zerodf_taylor_fwd": (C.c_int, [_P, _P, _L, _L, _I, _I, _I, C.c_double, _P, _I, _P, _P, _P]),
    "dsa_zerodf_taylor_bwd": (C.c_int, [_P, _P, _P, _L, _L, _I, _I, _I, C.c_double, _P, _I, _P, _P, _P]),
    "dsa_acorr_fwd": (C.c_int, [_P, _L, _I, _I, _I, _I, _P, _P]),
    "dsa_acorr_bwd": (C.c_int, [_P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "dsa_levdur_fwd": (C.c_int, [_P, _L, _I, _D, _I, _P, _P]),
    "dsa_levdur_bwd": (C.c_int, [_P, _P, _P, _L, _I, _D, _I, _P, _P]),
    "dsa_lpc_fwd": (C.c_int, [_P, _L, _I, _I, _D, _I, _P, _P, _P]),
    "dsa_lpc_bwd": (C.c_int, [_P, _P, _P, _L, _I, _I, _D, _I, _P, _P]),
    "dsa_frame_window_lpc_fwd": (C.c_int, [_P, _L, _L, _I, _I, _P, _I, _I, _I, _D, _I, _P, _P, _P]),
    "dsa_frame_window_lpc_bwd": (C.c_int, [_P, _P, _L, _L, _I, _I, _P, _I, _I, _I, _D, _I, _P, _P]),
}


def load():
    """Load the library (after torch, so both share one HIP runtime) and bind signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  diffsptk_amd has no CPU fallback."
        )
    import torch  # noqa: F401  (loads torch's libamdhip64.so first: one runtime per process)

    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise BackendError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().dsa_last_error().decode(errors="replace")
        raise BackendError(f"{what or 'diffsptk_amd call'} failed (status {rc}): {msg}")


def last_kernel() -> str:
    return load().dsa_last_kernel().decode()
