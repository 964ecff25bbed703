mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -k "rows_gemm or untuned" 2>&1 | tail -15 > gpurun_out/r4d/pytest.txt
timeout 300 python tools/time_48k.py > gpurun_out/r4d/time48k.txt 2>&1
cat gpurun_out/r4d/pytest.txt gpurun_out/r4d/time48k.txt
