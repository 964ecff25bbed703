#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_synth.py -x -q 2>&1 | tail -12 > gpurun_out/fir_test.log
cat gpurun_out/fir_test.log
timeout 600 python tools/bench_rows2.py 2>&1 | tail -5 > gpurun_out/rows2g.log
cat gpurun_out/rows2g.log
