#!/bin/bash
# kernel trace of one script: tools/gpu_trace.sh <script.py> <name>
R=$PWD; OUT=$R/gpurun_out/$2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/$1 > $OUT/trace.log 2>&1
cd $R
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_db_summary.py $DB > $OUT/kernel_trace.txt
rm -rf $OUT/trace
head -40 $OUT/kernel_trace.txt
