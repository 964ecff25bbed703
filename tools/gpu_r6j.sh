#!/bin/bash
mkdir -p gpurun_out/r6j
timeout 900 python tools/check_big_newton.py 2>&1 | tee gpurun_out/r6j/check.txt | tail -4
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
