#!/bin/bash
# round 6, first lease: the whole GPU suite on the purged library, then the cost of the purge against the round-5 library
mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -x -q > gpurun_out/r6a/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r6a/pytest.txt
tail -15 gpurun_out/r6a/pytest.txt
bash tools/ab_libs.sh r6a/purge_ab 3 tools/time_purge_ab.py build/lib_r5.so product
