#!/bin/bash
# Library variants for the reduction of the stale-result effect: tools/hazard/build_variants.sh [name:flags ...]
# Each variant recompiles csrc/mcep_mfma.hip with its -D flags and links it with the objects of the regular build
# (diffsptk_amd/lib/obj) into build/hz/lib_<name>.so.  Without arguments: the matrix of round 5.
cd "$(dirname "$0")/../.."
mkdir -p build/hz
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -mcode-object-version=5 -Wno-unused-value -ffp-contract=on"
if [ $# -eq 0 ]; then
  set -- "sc:" "pk31:-DDSA_FUSED_PK_MASK=31" "pk31_one:-DDSA_FUSED_PK_MASK=31 -DDSA_MCEP_ABL_ONEWAVE" \
     "pk1:-DDSA_FUSED_PK_MASK=1" "pk2:-DDSA_FUSED_PK_MASK=2" "pk4:-DDSA_FUSED_PK_MASK=4" "pk8:-DDSA_FUSED_PK_MASK=8" "pk16:-DDSA_FUSED_PK_MASK=16" \
     "pk31_d1:-DDSA_FUSED_PK_MASK=31 -DDSA_FUSED_DBG=1" "pk31_d2:-DDSA_FUSED_PK_MASK=31 -DDSA_FUSED_DBG=2" "pk31_d3:-DDSA_FUSED_PK_MASK=31 -DDSA_FUSED_DBG=3" \
     "pk31_d4:-DDSA_FUSED_PK_MASK=31 -DDSA_FUSED_DBG=4" "pk31_valu:-DDSA_FUSED_PK_MASK=31 -DDSA_MCEP_SOLVE_VALU" "chk:-DDSA_FUSED_DBG=8"
fi
build_one() {
  name=${1%%:*}; flags=${1#*:}
  hipcc $FL $flags -c diffsptk_amd/csrc/mcep_mfma.hip -o build/hz/$name.o 2> build/hz/$name.err || { echo "FAILED $name"; tail -5 build/hz/$name.err; return 1; }
  OBJS=""
  for o in diffsptk_amd/lib/obj/*.o; do
    [ "$(basename $o)" = "mcep_mfma.hip.o" ] && OBJS="$OBJS build/hz/$name.o" || OBJS="$OBJS $o"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o build/hz/lib_$name.so $OBJS 2>> build/hz/$name.err && echo "built $name ($flags)"
}
for v in "$@"; do build_one "$v" & 
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
