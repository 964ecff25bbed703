mkdir -p gpurun_out/r4j
timeout 1200 python -m pytest tests/test_gpu_synth.py tests/test_gpu_configs.py -q -x 2>&1 | tail -6 > gpurun_out/r4j/pytest.txt
python tools/time_mgcep.py > gpurun_out/r4j/mgcep.txt 2>&1
timeout 300 python tools/time_48k.py 2>&1 | head -3 >> gpurun_out/r4j/mgcep.txt
cat gpurun_out/r4j/pytest.txt gpurun_out/r4j/mgcep.txt
