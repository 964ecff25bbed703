#!/bin/bash
# round 6: the packed STFT kernel with larger workgroups (adjacent passes on one CU) -- timing + fetched bytes; the list of counters
mkdir -p gpurun_out/r6e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6e
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCC_EA0?_[A-Z0-9_]+|SQ_WAIT[A-Z_]*|SQ_INST_LEVEL[A-Z_]*|SQ_INSTS_VMEM[A-Z_]*|SQ_ACTIVE_INST[A-Z_]*|SQ_INST_CYCLES[A-Z_]*|TCP_PENDING[A-Z_]*|TCP_TCC[A-Z_]*|TCP_TA[A-Z_]*|TCC_[A-Z0-9_]*STALL[A-Z0-9_]*|TCC_HIT[A-Z_]*|TCC_MISS[A-Z_]*|TCC_WRITE[A-Z_]*|TCC_READ[A-Z_]*|TCC_REQ[A-Z_]*|TA_BUSY[A-Z_]*|TCC_NORMAL_WRITEBACK|TCC_ALL_TC_OP_WB_WRITEBACK)\b" | sort -u | tr '\n' ' ' > $OUT/counters.txt
cat $OUT/counters.txt; echo
for r in 1 2 3; do
for v in 2 8 9 10; do
  echo "DSA_STFT_PK=$v: $(DSA_STFT_PK=$v python tools/time_stft_all.py 2>/dev/null | tail -1)"
done
done | tee $OUT/timing.txt
cd /tmp; export TMPDIR=/tmp
for v in 2 9; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    DSA_STFT_PK=$v timeout 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/v$v/p$i -o p$i --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_stft_only.py > $OUT/v$v.p$i.log 2>&1
  done
  echo "== DSA_STFT_PK=$v" >> $OUT/summary.txt
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/v$v >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
