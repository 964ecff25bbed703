#!/bin/bash
mkdir -p gpurun_out/r6b
python tools/check_batch_invariance.py > gpurun_out/r6b/batch_invariance.txt 2>&1
cat gpurun_out/r6b/batch_invariance.txt | tail -30
python -m pytest tests -m gpu -q > gpurun_out/r6b/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r6b/pytest.txt
tail -8 gpurun_out/r6b/pytest.txt
python bench.py > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err; tail -3 gpurun_out/r6b/bench.err; head -c 3000 gpurun_out/r6b/bench.json
