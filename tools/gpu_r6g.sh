#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $OUT
for m in 1 2; do
for s in 2 1; do
  DSA_OVERLAP_MODE=$m timeout 300 python bench.py --streams $s --no-configs --no-cpu-baseline > $OUT/bench_m${m}_s$s.json 2> $OUT/err.txt
  python - <<PY
import json
d=json.load(open("$OUT/bench_m${m}_s$s.json"))
print("mode $m streams $s: ms_per_step", d["ms_per_step"], "single", (d.get("single_stream") or {}).get("ms_per_step"))
PY
done; done
cd /tmp; export TMPDIR=/tmp
for m in 1 2; do
  DSA_OVERLAP_MODE=$m timeout 200 rocprofv3 --kernel-trace -d $OUT/tr$m -o tr --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_fused_streams.py > $OUT/tr$m.log 2>&1
  echo "== mode $m"; python $GRAFT_REPO_ROOT/tools/trace_overlap.py $OUT/tr$m
done
OVERLAP=0 timeout 200 rocprofv3 --kernel-trace -d $OUT/tr0 -o tr --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_fused_streams.py > $OUT/tr0.log 2>&1
echo "== no flag, two streams"; python $GRAFT_REPO_ROOT/tools/trace_overlap.py $OUT/tr0
