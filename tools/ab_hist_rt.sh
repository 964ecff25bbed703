for v in 1 0 1 0; do
DSA_MCEP_HIST_RT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import json
d=json.load(open('bench_detail.json'))['configs']
print("HIST_RT=$v", {k:{kk:round(vv,4) for kk,vv in v.items() if kk.startswith('ms')} for k,v in d.items() if k.startswith('config3')})
PY
done
