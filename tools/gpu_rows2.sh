#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_synth.py -x -q 2>&1 | tail -5 > gpurun_out/mgcep_test.log
cat gpurun_out/mgcep_test.log
timeout 600 python tools/bench_rows2.py 2>&1 | tail -16 > gpurun_out/rows2f.log
cat gpurun_out/rows2f.log
