#!/bin/bash
mkdir -p gpurun_out/r6c
timeout 600 python -m pytest tests/test_gpu_cross_stream.py tests/test_gpu_fused_mcep.py -m gpu -q -x > gpurun_out/r6c/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r6c/pytest.txt
tail -3 gpurun_out/r6c/pytest.txt
for s in 2 1 2 1; do
  timeout 300 python bench.py --streams $s --no-configs --no-cpu-baseline > gpurun_out/r6c/bench_s$s.json 2> gpurun_out/r6c/bench_s$s.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r6c/bench_s$s.json"))
print("streams $s: ms_per_step", d["ms_per_step"], "value", d["value"], "avg_launch", d["roofline"]["avg_launch_ms"], "single", d.get("single_stream"))
PY
done
for s in 2 1 2 1; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --streams $s --no-configs --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver cmd streams $s:', d['ms_per_step'], d['value'], d.get('single_stream'))"
done
