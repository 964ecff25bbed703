#!/bin/bash
# full GPU suite + randomised sweeps of the 48 kHz kernels + the driver's bench line
mkdir -p gpurun_out/r6z
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r6z/pytest.txt
for s in 1 2 3; do timeout 300 python tools/fuzz_big.py $s 60 2>&1 | tail -40; done | tee gpurun_out/r6z/fuzz_big.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs 2>/dev/null | head -c 2500 | tee gpurun_out/r6z/bench.json
