mkdir -p gpurun_out/r4p
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r4p/pytest.txt
cat gpurun_out/r4p/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4p/bench_driver.json 2> gpurun_out/r4p/bench_driver.err
cp bench_detail.json gpurun_out/r4p/bench_detail_driver.json
timeout 300 python tools/time_48k.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4p/time48k.txt
bash tools/gpu_trace.sh tools/run_48k_only.py r4p_48k > /dev/null 2>&1
cat gpurun_out/r4p/bench_driver.json | cut -c1-1200; echo; cat gpurun_out/r4p/time48k.txt; head -8 gpurun_out/r4p_48k/kernel_trace.txt
