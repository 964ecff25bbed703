#!/bin/bash
# On the GPU box: every variant of build/hz/ through tools/hazard/hazard_check.py (the product library is put back afterwards).
# usage: tools/hazard/run_matrix.sh <out-name> [variant ...]
OUT=gpurun_out/$1; shift
mkdir -p gpurun_out
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
VARS="$@"
[ -z "$VARS" ] && VARS=$(ls build/hz/lib_*.so | sed 's/.*lib_\(.*\)\.so/\1/')
for v in $VARS; do
  cp build/hz/lib_$v.so diffsptk_amd/lib/libdiffsptk_amd.so
  timeout 120 python tools/hazard/hazard_check.py $v ${REPS:-3} ${NITER:-10} 2>&1 | grep -v amdgpu.ids
done > $OUT.txt 2>&1
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
cat $OUT.txt
