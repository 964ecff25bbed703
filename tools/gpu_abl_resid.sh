#!/bin/bash
# Ablations of mcep_resid_mfma_kernel on one box: which part of a chunk costs what (results are garbage in the ablated builds)
mkdir -p gpurun_out/abl
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
for v in 0 1 2 4 8 16 3 7 15 31; do
  cp build/librg_abl$v.so diffsptk_amd/lib/libdiffsptk_amd.so
  echo "RG_ABL=$v: $(ABL_ONLY=1 python tools/time_resid.py 2>/dev/null | head -2 | tr '\n' ' ')"
done > gpurun_out/abl/resid.txt 2>&1
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
cat gpurun_out/abl/resid.txt
