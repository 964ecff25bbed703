#!/bin/bash
# A/B of builds of the library on ONE box: tools/gpu_ab_lib.sh <script.py> build/libA.so build/libB.so ... (three rounds, alternating)
mkdir -p gpurun_out
S=$1; shift
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
for r in 1 2 3; do
  for v in "$@"; do
    cp $v diffsptk_amd/lib/libdiffsptk_amd.so
    python $S $v 2>/dev/null | tail -1
  done
done > gpurun_out/ab_lib.log 2>&1
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
cat gpurun_out/ab_lib.log
