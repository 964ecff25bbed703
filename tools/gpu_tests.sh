# full GPU suite + default bench (run via gpurun)
mkdir -p gpurun_out/t
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/t/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/t/bench.json 2> gpurun_out/t/bench.err
cat gpurun_out/t/pytest.txt; python -c "
import json
d=json.loads(open('gpurun_out/t/bench.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline_stft']['avg_launch_ms'], d['roofline_stft']['frac'])"
