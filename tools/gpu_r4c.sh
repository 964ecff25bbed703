mkdir -p gpurun_out/r4c
python tools/dbg_fused.py 2>&1 | grep "bad frames" > gpurun_out/r4c/dbg.txt
python tools/dbg_fused4.py >> gpurun_out/r4c/dbg.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_fused_mcep.py -q -x 2>&1 | tail -5 > gpurun_out/r4c/pytest.txt
timeout 300 python tools/time_fused_mcep.py > gpurun_out/r4c/time.txt 2>&1
cat gpurun_out/r4c/dbg.txt gpurun_out/r4c/pytest.txt gpurun_out/r4c/time.txt
