#!/bin/bash
# mel-cepstral kernels: parity tests (forward, gradients, bench-size float64 comparisons), then the headline + config-3 timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "grad or backward or config or reproducible or mcep or mgcep" 2>&1 | tail -4
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fwd', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], 'stft', d['roofline_stft']['avg_launch_ms'])
for k in ('config3_fwdbwd_batch256','config3_fwdbwd_batch1024'):
    c=d['configs'][k]; print(k, round(c['ms_fwd_bwd'],4), 'mcep_bwd', round(c['ms_mcep_bwd'],4), 'stft_bwd', round(c['ms_stft_bwd'],4))"
