#!/bin/bash
# PMC passes over a standalone binary: tools/pmc_bin.sh <outdir> <binary> [args]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT; BIN=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o p$i --output-format csv -- $BIN "$@" > $OUT/p$i.log 2>&1
done
