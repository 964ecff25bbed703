mkdir -p gpurun_out/r4b
timeout 600 python -m pytest tests/test_gpu_fused_mcep.py -q -x 2>&1 | tail -30 > gpurun_out/r4b/pytest.txt
timeout 300 python tools/time_fused_mcep.py > gpurun_out/r4b/time.txt 2>&1
cat gpurun_out/r4b/pytest.txt gpurun_out/r4b/time.txt
