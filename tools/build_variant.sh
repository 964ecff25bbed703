#!/bin/bash
# One library variant for A/B runs: tools/build_variant.sh <out.so> <source.hip> <extra -D flags...>   (the other objects come
# from the last regular build in diffsptk_amd/lib/obj)
OUT=$1; SRC=$2; shift 2
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mcode-object-version=5 -Wno-unused-value -ffp-contract=on"
EXTRA=""
[ "$SRC" = "stft.hip" ] && EXTRA="-Xclang -target-feature -Xclang -packed-fp32-ops"
hipcc $FL $EXTRA "$@" -c diffsptk_amd/csrc/$SRC -o build/variant_$(basename $OUT).o 2>/dev/null || exit 1
OBJS=""
for o in diffsptk_amd/lib/obj/*.o; do
  [ "$(basename $o)" = "$SRC.o" ] && OBJS="$OBJS build/variant_$(basename $OUT).o" || OBJS="$OBJS $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o $OUT $OBJS 2>/dev/null && echo built $OUT
