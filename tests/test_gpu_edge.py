"""Tensors beyond 2^31 elements ("maximum sizes"; DESIGN.md 2: sizes are bounded by the grid, not by 32-bit element offsets).

A batch whose spectrogram has more than 2^31 floats -- 42 000 utterances x 1 s at 16 kHz (8.4 M frames, 8 GiB), 10 600 at 48 kHz / fft 2048 --
against the same utterances alone in a small batch: the utterances at the ends and on both sides of the 2^31-element boundary must come
out BITWISE equal (every kernel of the path is batch-invariant), through the reference's module API: STFT, mcep(stft(x)), fuse(stft, mcep),
fuse(frame, window, lpc), the 48 kHz STFT + one-launch Newton kernel; the gradient of mcep(stft(x)) within the gradient tolerance of the
other tests (above ops.MCEP_HIST_RT_MAX_BYTES the backward recomputes the rt rows: another kernel) and exactly zero elsewhere.
The checks themselves live in tools/check_large_sizes.py (also a command-line report).
"""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_batches_beyond_2_31_elements_match_small_batches(capsys):
    free, _total = torch.cuda.mem_get_info(0)
    if free < 64 * 2 ** 30:
        pytest.skip("needs 64 GiB of free device memory")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_large_sizes.py")
    spec = importlib.util.spec_from_file_location("check_large_sizes", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.run_all()
    out = capsys.readouterr().out
    assert bad == 0, out
    assert out.count("bitwise equal to the small batch: True") >= 7, out


def test_empty_batches_and_shortest_inputs(capsys):
    """The reference's ATen ops accept a zero-size batch and any T >= 1: every module of the path (and its consumers, forward and
    backward, fused or not, 16 kHz and 48 kHz set-ups) returns the reference's shape for batch 0 -- the C-ABI treats a zero count as a
    no-op BEFORE it looks at the pointers (empty tensors have no storage) -- and T = 1, one period, one period + 1, frame_length - 1
    give finite outputs and gradients, fuse(stft, mcep) equal to mcep(stft(x)) bit for bit (tools/check_empty_inputs.py)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_empty_inputs.py")
    spec = importlib.util.spec_from_file_location("check_empty_inputs", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.run_all()
    out = capsys.readouterr().out
    assert bad == 0, out
    assert "FAILED" not in out and "MISMATCH" not in out, out


def test_input_forms_the_reference_accepts(capsys):
    """Non-contiguous views (strided, transposed, expanded, offset), no / three leading dims, non-contiguous spectrograms and
    cotangents: bit-identical to the contiguous copy; float64 modules; fftr with inputs shorter / longer than fft_length against
    torch.fft.rfft(x, n) on the host (tools/check_input_forms.py)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_input_forms.py")
    spec = importlib.util.spec_from_file_location("check_input_forms", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.run_all()
    out = capsys.readouterr().out
    assert bad == 0, out
    assert "FAILED" not in out and "MISMATCH" not in out and out.count(": ok") >= 50, out
