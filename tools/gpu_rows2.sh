#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_synth.py -x -q 2>&1 | tail -12 > gpurun_out/mixed_test.log
cat gpurun_out/mixed_test.log
