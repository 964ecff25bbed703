mkdir -p gpurun_out/r4k
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r4k/pytest.txt
cat gpurun_out/r4k/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4k/bench_driver.json 2> gpurun_out/r4k/bench_driver.err
cp bench_detail.json gpurun_out/r4k/bench_detail_driver.json
python bench.py > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err
cp bench_detail.json gpurun_out/r4k/bench_detail.json
python bench.py --path fused --no-configs --no-cpu-baseline > gpurun_out/r4k/bench_fused.json 2> gpurun_out/r4k/bench_fused.err
python bench.py --global-batch 8192 --no-configs --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r4k/bench_strong8192.json 2> gpurun_out/r4k/bench_strong.err
cat gpurun_out/r4k/bench_driver.json; echo; cat gpurun_out/r4k/bench_fused.json | cut -c1-900; echo; cat gpurun_out/r4k/bench_strong8192.json | cut -c1-700
