#!/bin/bash
# A/B of two builds of the library over the headline bench on ONE box: tools/gpu_ab_lib.sh build/libA.so build/libB.so
mkdir -p gpurun_out
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
for v in $1 $2 $1 $2 $1 $2; do
  cp $v diffsptk_amd/lib/libdiffsptk_amd.so
  python bench.py --no-configs --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']/1e8,4), round(d['ms_per_step'],4), 'mcep', round(d['roofline']['avg_launch_ms'],4), 'b2b', round(d['roofline']['back_to_back_ms'],4), 'stft', round(d['roofline_stft']['avg_launch_ms'],4))"
done > gpurun_out/ab_lib.log 2>&1
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
cat gpurun_out/ab_lib.log
