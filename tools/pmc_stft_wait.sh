#!/bin/bash
# What the packed STFT kernels wait for: SQ wait / active / in-flight counters, L2 <-> fabric request counters (partial-line writes,
# stalls, queue levels), L1 latency counters -- one counter group per rocprofv3 run, never with other trace domains.
# usage: tools/pmc_stft_wait.sh <outdir-under-gpurun_out> <script.py>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; S=$GRAFT_REPO_ROOT/$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS" \
           "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL" \
           "TCC_HIT TCC_MISS TCC_TAG_STALL TCC_IB_STALL" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_REQ TCC_WRITE" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ" \
           "TCP_TCC_WRITE_REQ GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o p$i --output-format csv -- python $S > $OUT/p$i.log 2>&1
  echo "pass $i ($PMC) rc=$?" >> $OUT/passes.log
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/passes.log; cat $OUT/summary.txt
