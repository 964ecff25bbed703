mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_gpu_synth.py -q -x -k "thsolve" 2>&1 | tail -4 > gpurun_out/r4i/pytest.txt
timeout 300 python tools/time_48k.py 2>&1 | head -3 > gpurun_out/r4i/time48k.txt
cat gpurun_out/r4i/pytest.txt gpurun_out/r4i/time48k.txt
