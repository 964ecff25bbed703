# round-4 first call: full GPU suite + the driver's bench command (does the line parse?)
mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4a/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4a/bench_driver.json 2> gpurun_out/r4a/bench_driver.err
cat gpurun_out/r4a/pytest.txt; wc -c gpurun_out/r4a/bench_driver.json; cat gpurun_out/r4a/bench_driver.json; tail -3 gpurun_out/r4a/bench_driver.err
