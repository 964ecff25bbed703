#!/bin/bash
# Counter passes over the issue-rate calibration (tools/bench_issue.cpp): instructions / GRBM_GUI_ACTIVE is the issue rate
# independent of the clock; GRBM_GUI_ACTIVE / wall time is the clock.  usage: tools/issue_pmc.sh <outdir-under-gpurun_out> [filter] [iters]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
BIN=$GRAFT_REPO_ROOT/build/bench_issue
cd /tmp && export TMPDIR=/tmp
$BIN "${2:-}" ${3:-2000} > $OUT/plain.txt 2>&1
i=0
for PMC in "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o p$i --output-format csv -- $BIN "${2:-}" ${3:-2000} > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?" >> $OUT/summary.log
done
python $GRAFT_REPO_ROOT/tools/issue_pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
