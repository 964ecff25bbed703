#!/bin/bash
mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r6h/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r6h/pytest.txt
tail -8 gpurun_out/r6h/pytest.txt
for s in 2 1; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --streams $s --no-configs --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver cmd streams $s:', d['ms_per_step'], d['value'], d.get('single_stream'))"
done
