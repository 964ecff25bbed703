mkdir -p gpurun_out/s
./build/bench_stft_pk 1024 2>&1 | grep -E "^B=|DIRECT" 
timeout 600 python -m pytest tests -m gpu -q -x -k "stft or config or bitwise" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline_stft']; print('bench', d['value'], r['avg_launch_ms'], r['frac'], r['back_to_back_ms'])"
