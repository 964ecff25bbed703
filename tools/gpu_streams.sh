#!/bin/bash
mkdir -p gpurun_out
for cfg in "--streams 1" "--streams 2" "--streams 2 --record-every 8" "--streams 3 --record-every 8" "--streams 1" "--streams 2 --record-every 8"; do
  python bench.py --no-configs --no-cpu-baseline $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', round(d['value']/1e8,4), round(d['ms_per_step'],4), 'mcep', round(d['roofline']['avg_launch_ms'],4), 'stft', round(d['roofline_stft']['avg_launch_ms'],4))"
done > gpurun_out/streams.log 2>&1
cat gpurun_out/streams.log
