#!/bin/bash
# A/B of one environment knob over the headline bench: tools/gpu_ab.sh <ENV_NAME> <value A> <value B>
mkdir -p gpurun_out
for v in $2 $3 $2 $3; do
  env $1=$v python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1=$v', round(d['value']/1e8,4), round(d['ms_per_step'],4), 'mcep', round(d['roofline']['avg_launch_ms'],4), 'stft', round(d['roofline_stft']['avg_launch_ms'],4))"
done > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "mcep or config or reproducible" 2>&1 | tail -3
