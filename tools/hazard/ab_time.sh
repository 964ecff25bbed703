#!/bin/bash
# On the GPU box: time the fused STFT -> mel-cepstrum launch under several library variants, alternating, three rounds.
# usage: tools/hazard/ab_time.sh <out-name> <variant.so> ...   ("product" = the library in place)
OUT=gpurun_out/$1.txt; shift
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
for r in 1 2 3; do
  for v in "$@"; do
    [ "$v" = "product" ] && cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so || cp $v diffsptk_amd/lib/libdiffsptk_amd.so
    echo "$v: $(python tools/time_fused_mcep.py 2>/dev/null | tail -1)"
  done
done > $OUT 2>&1
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
cat $OUT
