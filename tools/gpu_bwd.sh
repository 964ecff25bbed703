mkdir -p gpurun_out/bwd
for a in 0 1 2 3 4 7; do echo "ABL=$a"; DSA_MCEP_BWD_ABL=$a python tools/bench_fwdbwd.py 1024 2>&1 | sed -n 2,3p; done
timeout 600 python -m pytest tests -m gpu -q -x -k "mcep or config3 or bitwise or many_launches" 2>&1 | tail -5
