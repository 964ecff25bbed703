#!/bin/bash
mkdir -p gpurun_out/r6p
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
cp build/lib_bigstamps.so diffsptk_amd/lib/libdiffsptk_amd.so
python tools/big_stamps.py 2>&1 | tee gpurun_out/r6p/stamps.txt
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
