mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_synth.py -q -x -k "rows_gemm or untuned or thsolve or freqt or mgcep or 48" 2>&1 | tail -5 > gpurun_out/r4l/pytest.txt
timeout 300 python tools/time_48k.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4l/time48k.txt
for nt in 2 4; do echo "DSA_ROWS_GEMM_NT=$nt" >> gpurun_out/r4l/time48k.txt; DSA_ROWS_GEMM_NT=$nt timeout 300 python tools/time_48k.py 2>&1 | grep -v amdgpu.ids | head -2 >> gpurun_out/r4l/time48k.txt; done
cat gpurun_out/r4l/pytest.txt gpurun_out/r4l/time48k.txt
bash tools/gpu_trace.sh tools/run_48k_only.py r4l_48k | head -10
