mkdir -p gpurun_out/r4g
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4g/pytest.txt
cat gpurun_out/r4g/pytest.txt
bash tools/gpu_bench.sh r4g > gpurun_out/r4g/gpu_bench.log 2>&1
tail -60 gpurun_out/r4g/gpu_bench.log | cut -c1-400
