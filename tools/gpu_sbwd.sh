#!/bin/bash
# packed STFT backward: parity against the two-kernel path, then kernel times (rocprofv3) for a few knob settings
DSA_STFT_BWD_PK=0 timeout 200 python tools/ab_stft_bwd.py old 2>&1 | tail -1
timeout 200 python tools/ab_stft_bwd.py new old 2>&1 | tail -20 | awk '{print $1, $8}' | tr '\n' ' '; echo
for mr in 6 3 12; do
  export DSA_STFT_BWD_MINRUN=$mr
  bash tools/gpu_trace.sh tools/run_stft_bwd_only.py sb_mr$mr 2>&1 | grep "bwd_pk" | cut -c1-130
done
