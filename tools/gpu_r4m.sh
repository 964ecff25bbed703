mkdir -p gpurun_out/r4m
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_synth.py -q -x -k "rows_gemm or untuned or thsolve or freqt or mgcep or 48 or newton" 2>&1 | tail -5 > gpurun_out/r4m/pytest.txt
timeout 300 python tools/time_48k.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4m/time48k.txt
echo "DSA_MCEP_RESID=0" >> gpurun_out/r4m/time48k.txt; DSA_MCEP_RESID=0 timeout 300 python tools/time_48k.py 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/r4m/time48k.txt
cat gpurun_out/r4m/pytest.txt gpurun_out/r4m/time48k.txt
bash tools/gpu_trace.sh tools/run_48k_only.py r4m_48k | head -10
