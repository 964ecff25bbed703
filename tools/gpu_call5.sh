mkdir -p gpurun_out/c6
./build/bench_stft_pk 1024 > gpurun_out/c6/pk_abl.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "stft or config" 2>&1 | tail -15 > gpurun_out/c6/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err
DSA_STFT_PK=1 python bench.py --no-cpu-baseline > gpurun_out/c6/bench_pk1.json 2> gpurun_out/c6/bench_pk1.err
cat gpurun_out/c6/pk_abl.txt gpurun_out/c6/pytest.txt; python -c "
import json
for n in ('bench','bench_pk1'):
    d=json.loads(open('gpurun_out/c6/%s.json'%n).read().strip().split('\n')[-1]); print(n, d['value'], d['ms_per_step'], d['roofline_stft']['avg_launch_ms'], d['roofline_stft']['frac'])"
