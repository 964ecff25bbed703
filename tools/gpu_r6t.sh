#!/bin/bash
# in-context (between mel-cepstral launches) and back-to-back time of the packed STFT kernel's variants, as bench.py measures them
mkdir -p gpurun_out/r6t
for rep in 1 2; do
for v in "2 0" "8 0" "9 0" "10 0" "7 0" "2 2" "2 4" "9 2" "3 0" "6 0"; do
  set -- $v
  DSA_STFT_PK=$1 DSA_STFT_RUN=$2 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline_stft']; print('PK=$1 RUN=$2: in-context', round(r['avg_launch_ms']*1e3,1), 'b2b', round(r['back_to_back_ms']*1e3,1), 'kernel', r['kernel'])"
done
done | tee gpurun_out/r6t/stft_variants_in_context.txt
