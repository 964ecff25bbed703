#!/bin/bash
# phase stamps of the mel-cepstral forward kernel (wave 0, steady state): two waves per SIMD and one wave alone
OUT=gpurun_out/r6i; mkdir -p $OUT build
FL="--offload-arch=gfx950 -O3 -std=c++17 -mcode-object-version=5 -Wno-unused-value -ffp-contract=on -Xclang -target-feature -Xclang -packed-fp32-ops -DDSA_MCEP_TIMING -Iinclude"
hipcc $FL tools/bench_mcep.cpp -o build/bench_mcep 2> $OUT/build.log || tail -5 $OUT/build.log
hipcc $FL -DDSA_MCEP_ABL_ONEWAVE tools/bench_mcep.cpp -o build/bench_mcep_one 2>> $OUT/build.log
for i in 1 2 3; do build/bench_mcep 204800 40; done > $OUT/stamps_two_waves.txt 2>&1
for i in 1 2; do build/bench_mcep_one 204800 20; done > $OUT/stamps_one_wave.txt 2>&1
cat $OUT/stamps_two_waves.txt; echo ----; cat $OUT/stamps_one_wave.txt
