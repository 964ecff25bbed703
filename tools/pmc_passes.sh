#!/bin/bash
# Counter passes for the bench kernels (run on the GPU box; PMC only with --kernel-trace).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# forward kernels: the headline command without its sub-benchmarks (every launch covers 204 800 frames);
# backward kernels: a few forward + backward steps at the same size (2nd argument "bwd")
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --no-configs --path two-kernel"
if [ "${2:-fwd}" = "bwd" ]; then CMD="python $GRAFT_REPO_ROOT/tools/run_mcep_bwd_only.py"; fi
if [ "${2:-fwd}" = "fused" ]; then CMD="python $GRAFT_REPO_ROOT/tools/run_fused_only.py"; fi
if [ "${2:-fwd}" = "fusedmcep" ]; then CMD="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --no-configs --path fused"; fi
if [ "${2:-fwd}" = "lpc" ]; then CMD="python $GRAFT_REPO_ROOT/tools/run_lpc_only.py"; fi
if [ "${2:-fwd}" = "lpcbwd" ]; then CMD="python $GRAFT_REPO_ROOT/tools/run_lpc_bwd_only.py"; fi
if [ "${2:-fwd}" = "mgcep" ]; then CMD="python $GRAFT_REPO_ROOT/tools/run_mgcep_only.py"; fi
if [ "${2:-fwd}" = "k48" ]; then CMD="env B=512 N=3 python $GRAFT_REPO_ROOT/tools/run_48k_only.py"; fi
if [ "${2:-fwd}" = "mlsa" ]; then CMD="python $GRAFT_REPO_ROOT/tools/run_mlsa_multistage_only.py"; fi
if [ "${2:-fwd}" = "stftbwd" ]; then CMD="env N=6 SMALL=0 python $GRAFT_REPO_ROOT/tools/run_stft_bwd_only.py"; fi
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/p$i -o p$i --output-format csv -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?" >> $OUT/summary.log
done
find $OUT -name "*.csv" | head -20 >> $OUT/summary.log
