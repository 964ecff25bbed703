python tools/bench_fwdbwd.py 1024 2>&1 | sed -n 2,3p
python tools/bench_fwdbwd.py 256 2>&1 | sed -n 2,3p
timeout 600 python -m pytest tests -m gpu -q -x -k "mcep or config3 or bitwise or many_launches" 2>&1 | tail -3
