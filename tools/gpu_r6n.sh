#!/bin/bash
mkdir -p gpurun_out/r6n
for n in 10 1 3; do
  python tools/hazard/hazard_check.py pk31_safe 100 $n > gpurun_out/r6n/hc_n$n.txt 2>&1
  echo "n_iter=$n: launches $(grep -c 'bad frames' gpurun_out/r6n/hc_n$n.txt), launches with 0 bad frames $(grep -c 'bad frames 0 ' gpurun_out/r6n/hc_n$n.txt)"
done | tee gpurun_out/r6n/hazard_summary.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver cmd:', d['ms_per_step'], d['value'], 'two_streams', d.get('two_streams'), 'frac', d['roofline']['frac'])"
