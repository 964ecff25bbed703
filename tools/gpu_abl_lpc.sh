#!/bin/bash
# Ablations of frame_window_lpc24_mfma_kernel on one box (results are garbage in the ablated builds)
mkdir -p gpurun_out/abl
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
for v in 0 1 2 4 8 3 6 15; do
  cp build/liblpc_abl$v.so diffsptk_amd/lib/libdiffsptk_amd.so
  echo "LPC_ABL=$v: $(python tools/time_lpc.py 2>/dev/null | tail -1)"
done > gpurun_out/abl/lpc.txt 2>&1
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
cat gpurun_out/abl/lpc.txt
