#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fftcep or reproducible" 2>&1 | tail -15 > gpurun_out/fftcep_test.log
cat gpurun_out/fftcep_test.log
timeout 600 python tools/bench_rows2.py 2>&1 | head -5 > gpurun_out/rows2b.log
cat gpurun_out/rows2b.log
