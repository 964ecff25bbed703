#!/bin/bash
# A/B: the arrival counter at the end of the persistent mel-cepstral forward (2048 atomics on one address) against a fill launch per call
cp diffsptk_amd/lib/libdiffsptk_amd.so /tmp/lib_orig.so
for r in 1 2 3; do
  echo "clean scratch + arrival counter: $(python bench.py --no-configs --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])")"
  echo "fill launch, arrival counter:    $(DSA_CLEAN_SCRATCH=0 python bench.py --no-configs --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])")"
  cp build/libmm_noexit.so diffsptk_amd/lib/libdiffsptk_amd.so
  echo "fill launch, no arrival counter: $(DSA_CLEAN_SCRATCH=0 python bench.py --no-configs --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])")"
  cp /tmp/lib_orig.so diffsptk_amd/lib/libdiffsptk_amd.so
done
