#!/bin/bash
# fused STFT -> filter bank kernel: parity tests, then timings (gpurun_out/fused.log)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -25 > gpurun_out/fused_test.log
cat gpurun_out/fused_test.log
timeout 300 python tools/bench_fused.py 2>&1 | tail -12 > gpurun_out/fused.log
cat gpurun_out/fused.log
