# default bench (as the driver runs it) + kernel trace + PMC passes; outputs under gpurun_out/$1; ROUND (default r06) names the
# profile files the traffic record points at (copy gpurun_out/$1/pmc_<what>.txt to profiles/${ROUND}_pmc_<what>_$1.txt)
ROUND=${ROUND:-r06}
OUT=gpurun_out/$1
mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o trace -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs > $R/$OUT/trace.log 2>&1
cd $R
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_db_summary.py $DB > $OUT/kernel_trace.txt
bash tools/pmc_passes.sh $1/pmc_fwd fwd
bash tools/pmc_passes.sh $1/pmc_bwd bwd
bash tools/pmc_passes.sh $1/pmc_fused fused
bash tools/pmc_passes.sh $1/pmc_sbwd stftbwd
bash tools/pmc_passes.sh $1/pmc_lpc lpc
bash tools/pmc_passes.sh $1/pmc_lpcbwd lpcbwd
bash tools/pmc_passes.sh $1/pmc_fusedmcep fusedmcep
bash tools/pmc_passes.sh $1/pmc_k48 k48
python tools/pmc_summary.py $OUT/pmc_k48 > $OUT/pmc_k48.txt 2>&1
bash tools/gpu_trace.sh tools/run_stft_bwd_only.py $1/sbwd_trace > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pmc_fwd > $OUT/pmc_fwd.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_bwd > $OUT/pmc_bwd.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_fused > $OUT/pmc_fused.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_sbwd > $OUT/pmc_sbwd.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_lpc > $OUT/pmc_lpc.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_lpcbwd > $OUT/pmc_lpcbwd.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_fusedmcep > $OUT/pmc_fusedmcep.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_fwd,$OUT/pmc_fusedmcep,$OUT/pmc_bwd,$OUT/pmc_fused,$OUT/pmc_sbwd,$OUT/pmc_lpc,$OUT/pmc_lpcbwd "profiles/${ROUND}_pmc_{fwd,fusedmcep,bwd,fused,sbwd,lpc,lpcbwd}_$1.txt (rocprofv3 --pmc passes of tools/pmc_passes.sh: bench.py --no-configs [--path fused], tools/run_mcep_bwd_only.py, tools/run_fused_only.py, tools/run_stft_bwd_only.py, tools/run_lpc_only.py and tools/run_lpc_bwd_only.py, 204800 frames per launch)" > $OUT/pmc_traffic.json
rm -rf $OUT/trace/*/*.db.tmp
head -c 1500 $OUT/bench_driver.json; echo; tail -3 $OUT/bench.err; head -30 $OUT/kernel_trace.txt; cat $OUT/pmc_traffic.json | head -80
