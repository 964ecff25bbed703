mkdir -p gpurun_out/r4h
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -k "rows_gemm or untuned" 2>&1 | tail -5 > gpurun_out/r4h/pytest.txt
timeout 300 python tools/time_48k.py > gpurun_out/r4h/time48k.txt 2>&1
cat gpurun_out/r4h/pytest.txt gpurun_out/r4h/time48k.txt
bash tools/gpu_trace.sh tools/run_48k_only.py r4_48k_c | head -8
