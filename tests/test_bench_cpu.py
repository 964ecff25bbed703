"""CPU tests of the measurement plumbing: the one JSON line bench.py prints, and that the instruction-mix tool only reads
the product library (the round-3 record was unparseable and tools/isa_mix.py rewrote the .so it measured)."""
import hashlib
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _canned(bench):
    note = "n" * 1500   # the long prose that used to ride in the line
    roof = {"kernel": "stft512_mcep_fused_fwd", "bound": "f32_datapath (vector ALU + float32 matrix instructions share it)",
            "achieved": 1272.123456789, "peak": 2457.6, "unit": "G datapath-cycles/s", "frac": 0.5176543, "traffic": 2.357e8,
            "avg_launch_ms": 0.5534, "note": note, "pmc": {"valu_busy": 0.55}, "datapath": {"x": list(range(200))}}
    return {
        "metric": "frames/sec STFT->mcep (fl=400 fp=80 nfft=512 M=24)", "value": 3.2612345678e8, "unit": "frames/s", "n_gpus": 1,
        "steps": 20, "warmup": 5, "ms_per_step": 0.628123456, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] per-GPU shard: STFT->mcep forward, 1024 utterances x 1 s @ 16 kHz per GPU "
                               "(204800 frames), alpha=0.42 n_iter=10; N=8 is the full 8192-utterance batch; features all-gathered "
                               "over RCCL when N>1", "utterances_per_gpu": 1024, "global_batch": 1024, "frames_per_step": 204800,
                   "parallelism": "dp1", "kernels": {"stft": "stft512_fwd", "mcep": "mcep_mfma_fwd"}, "arith": note},
        "roofline": roof,
        "roofline_stft": dict(roof, kernel="stft512_fwd", bound="hbm", unit="GB/s", peak=8000.0, achieved=3516.2, frac=0.4395),
        "configs": {"big": [note] * 12},
        "cpu_baseline": {"value": 40912.3, "unit": "frames/s", "cores": 128, "threads_best": 8, "kind": "port", "sample": "64 utterances x 1 s " + "s" * 300,
                         "by_threads": {str(i): {"value": 1.0 * i, "note": note} for i in (1, 8, 32, 128)}, "host": {"cpu": note}},
        "gpu_over_cpu": 7971.2, "cpu_baseline_c_oracle": {"value": 1.0, "note": note},
    }


def test_bench_line_is_compact_parseable_and_starts_with_metric(tmp_path, monkeypatch):
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_under_test")
    res = _canned(bench)
    assert len(json.dumps(res)) > 20000   # the full record is what broke the driver's parser
    line = bench.compact_line(res)
    assert len(line) < 4096 and "\n" not in line
    assert line.startswith('{"metric"')
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_stft", "cpu_baseline", "detail"):
        assert k in d, k
    assert d["value"] == pytest.approx(res["value"], rel=1e-4) and d["ms_per_step"] == pytest.approx(res["ms_per_step"], rel=1e-4)
    for r in ("roofline", "roofline_stft"):
        assert set(d[r]) == {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"}
        assert d[r]["frac"] == pytest.approx(d[r]["achieved"] / d[r]["peak"], rel=1e-3)
    assert set(d["cpu_baseline"]) == {"value", "unit", "cores", "threads_best", "kind", "sample"} and d["cpu_baseline"]["kind"] in ("port", "reference")
    assert "workload" in d["config"] and "model" not in d["config"]
    # emit(): the full record goes to the side file, stdout gets the compact line only
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    line2 = bench.emit(res)
    assert line2 == line
    full = json.load(open(tmp_path / bench.DETAIL_FILE))
    assert full["configs"] == res["configs"] and full["roofline"]["note"] == res["roofline"]["note"]


def test_bench_line_survives_oversized_texts():
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_under_test2")
    res = _canned(bench)
    res["config"]["workload"] = "w" * 6000
    res["cpu_baseline"]["sample"] = "s" * 6000
    line = bench.compact_line(res)
    assert len(line) < 4096 and line.startswith('{"metric"')
    json.loads(line)


def test_isa_mix_does_not_touch_the_library():
    lib = os.path.join(ROOT, "diffsptk_amd", "lib", "libdiffsptk_amd.so")
    if not os.path.exists(lib) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built library and llvm-objdump")
    before = (hashlib.md5(open(lib, "rb").read()).hexdigest(), os.stat(lib).st_mtime_ns, os.stat(lib).st_size)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_mix
    finally:
        sys.path.pop(0)
    ks = isa_mix.disassemble(lib)
    assert any("mcep_mfma_fwd_kernel_h" in k for k in ks)
    after = (hashlib.md5(open(lib, "rb").read()).hexdigest(), os.stat(lib).st_mtime_ns, os.stat(lib).st_size)
    assert before == after
