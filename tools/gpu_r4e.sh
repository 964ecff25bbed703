mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests/test_gpu_synth.py -q -x -k "thsolve" 2>&1 | tail -15 > gpurun_out/r4e/pytest.txt
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -k "rows_gemm or untuned" 2>&1 | tail -8 >> gpurun_out/r4e/pytest.txt
cat gpurun_out/r4e/pytest.txt
bash tools/gpu_trace.sh tools/run_48k_only.py r4_48k_b | head -14
