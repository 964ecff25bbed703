for r in 1 2; do for v in 2 3 4 5 6; do
  env DSA_STFT_PK=$v python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PK=$v', round(d['value']/1e8,4), round(d['ms_per_step'],4), 'mcep', round(d['roofline']['avg_launch_ms'],4), 'stft', round(d['roofline_stft']['avg_launch_ms'],4), 'b2b', round(d['roofline_stft']['back_to_back_ms'],4))"
done; done
