#!/bin/bash
# HBM bytes of the packed STFT kernel under both pass orders (round-robin / runs of 13): FETCH_SIZE and WRITE_SIZE in separate passes
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for R in 0 13; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    DSA_STFT_RUN=$R timeout 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/r$R/p$i -o p$i --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_stft_only.py > $OUT/r$R.p$i.log 2>&1
  done
  echo "== DSA_STFT_RUN=$R" >> $OUT/summary.txt
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/r$R >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
