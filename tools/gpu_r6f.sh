#!/bin/bash
mkdir -p gpurun_out/r6f
bash tools/pmc_stft_wait.sh r6f/stft512 tools/run_stft_only.py > gpurun_out/r6f/stft512.txt 2>&1
bash tools/pmc_stft_wait.sh r6f/stftbig tools/run_stft_big_only.py > gpurun_out/r6f/stftbig.txt 2>&1
tail -70 gpurun_out/r6f/stft512.txt
python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "speech" 2>&1 | tail -3
