#!/bin/bash
# soak of the final library: the GPU suite twice, randomised parity sweeps (old and new kernels), repeat-launch determinism, the failing-frame count of the fused launch
mkdir -p gpurun_out/soak
for i in 1 2; do python -m pytest tests -m gpu -x -q 2>&1 | tail -2; done | tee gpurun_out/soak/pytest.txt
for s in ${SOAK_SEEDS_A:-21 22 23}; do timeout 300 python tools/fuzz_parity.py $s 40 2>&1 | tail -30; done > gpurun_out/soak/fuzz_parity.txt 2>&1
for s in ${SOAK_SEEDS_B:-31 32}; do timeout 300 python tools/fuzz_big.py $s 60 2>&1 | grep -v "^kernel" | tail -16; done > gpurun_out/soak/fuzz_big.txt 2>&1
timeout 600 python tools/hazard/stress_determinism.py 60 > gpurun_out/soak/determinism.txt 2>&1
timeout 600 python tools/hazard/hazard_check.py product 100 > gpurun_out/soak/hazard_check.txt 2>&1
tail -3 gpurun_out/soak/fuzz_parity.txt; grep -c LARGE gpurun_out/soak/fuzz_big.txt; cat gpurun_out/soak/determinism.txt | tail -30; tail -5 gpurun_out/soak/hazard_check.txt
