# bench.py with one and with two streams, twice each (run via gpurun)
for s in 1 2 1 2; do
  python bench.py --streams $s --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('streams $s', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline_stft']['avg_launch_ms'])"
done
