#!/bin/bash
# On the GPU box: the determinism tests + the bench-size spectrogram check (hazard_check.py, N launches) under library variants.
# usage: tools/hazard/run_variant_tests.sh <launches> <variant.so> ...
N=$1; shift
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
for v in "$@"; do
  cp $v diffsptk_amd/lib/libdiffsptk_amd.so
  echo "== $v"
  timeout 500 python -m pytest tests/test_gpu_fused_mcep.py -m gpu -x -q 2>&1 | tail -1
  python tools/hazard/hazard_check.py $(basename $v .so) $N 2>&1 | grep -c "bad frames 0 " | sed "s/^/   launches with 0 bad frames (of $N): /"
  python tools/hazard/hazard_check.py $(basename $v .so) 10 3 2>&1 | grep -c "bad frames 0 " | sed "s/^/   n_iter = 3: launches with 0 bad frames (of 10): /"
done
cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
