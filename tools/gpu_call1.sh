mkdir -p gpurun_out/c1
./build/test_pk_asm > gpurun_out/c1/pk_asm.txt 2>&1
DSA_STFT_PK=1 ./build/bench_stft 1024 > gpurun_out/c1/bench_stft_pk1.txt 2>&1
DSA_STFT_PK=0 ./build/bench_stft 1024 > gpurun_out/c1/bench_stft_pk0.txt 2>&1
DSA_STFT_PK=1 ./build/bench_stft 64 > gpurun_out/c1/bench_stft_pk1_b64.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_configs.py 2>&1 | tail -30 > gpurun_out/c1/pytest_old.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -60 > gpurun_out/c1/pytest_new.txt
DSA_STFT_PK=1 python bench.py --no-cpu-baseline > gpurun_out/c1/bench_pk1.json 2> gpurun_out/c1/bench_pk1.err
DSA_STFT_PK=0 python bench.py --no-cpu-baseline > gpurun_out/c1/bench_pk0.json 2> gpurun_out/c1/bench_pk0.err
cat gpurun_out/c1/pk_asm.txt gpurun_out/c1/bench_stft_pk1.txt gpurun_out/c1/bench_stft_pk0.txt gpurun_out/c1/pytest_old.txt gpurun_out/c1/pytest_new.txt
