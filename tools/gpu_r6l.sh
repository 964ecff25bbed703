#!/bin/bash
for i in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default:', d['ms_per_step'], 'two_streams', d.get('two_streams'), 'frac', d['roofline']['frac'])"
done
timeout 300 build/repro_rot 20000 2>&1 | tee gpurun_out/repro_rot_r6.txt
