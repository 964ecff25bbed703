#!/bin/bash
# On the GPU box: one script under several library variants, alternating (the product library is put back afterwards).
# usage: tools/ab_libs.sh <out-name> <rounds> <script.py> <variant.so | product> ...      (last line of the script's output per run)
OUT=gpurun_out/$1.txt; R=$2; S=$3; shift 3
mkdir -p gpurun_out
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
for r in $(seq $R); do
  for v in "$@"; do
    [ "$v" = "product" ] && cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so || cp $v diffsptk_amd/lib/libdiffsptk_amd.so
    echo "$v: $(python $S 2>/dev/null | tail -${TAIL:-1} | tr '\n' ' ')"
  done
done > $OUT 2>&1
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
cat $OUT
