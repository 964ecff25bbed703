#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_synth.py -x -q 2>&1 | tail -15 > gpurun_out/fftrows_test.log
cat gpurun_out/fftrows_test.log
timeout 600 python tools/bench_rows2.py 2>&1 | tail -30 > gpurun_out/rows2.log
cat gpurun_out/rows2.log
