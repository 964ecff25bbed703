#!/bin/bash
mkdir -p gpurun_out/r6d
python -m pytest tests/test_gpu_stft_big.py tests/test_gpu_cross_stream.py -m gpu -q -x > gpurun_out/r6d/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r6d/pytest.txt
tail -25 gpurun_out/r6d/pytest.txt
python tools/time_stft_big.py 2>&1 | tee gpurun_out/r6d/time_big.txt
DSA_STFT_BIG=0 python tools/time_stft_big.py 2>&1 | tee gpurun_out/r6d/time_generic.txt
for s in 2 1; do
python bench.py --gpus 1 --steps 20 --warmup 5 --streams $s --no-configs --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver cmd streams $s:', d['ms_per_step'], d['value'], d.get('single_stream'))"
done
