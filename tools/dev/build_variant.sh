#!/bin/bash
# developer: build libganq_hip with extra -D flags for ONE source file into build_variants/<name>/libganq_hip.so
# usage: tools/dev/build_variant.sh <name> <source.hip> "<flags>"      (other objects are taken from ganq_amd/csrc/*.o)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; SRC=$2; FLAGS=$3
OUT=$ROOT/build_variants/$NAME
mkdir -p "$OUT"
make -C "$ROOT/ganq_amd/csrc" -j8 >/dev/null
/opt/rocm/bin/hipcc $FLAGS --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wall -Wno-unused-result -Wno-pass-failed -c "$ROOT/ganq_amd/csrc/$SRC" -o "$OUT/${SRC%.hip}.o"
OBJS=""
for o in "$ROOT"/ganq_amd/csrc/*.o; do
  b=$(basename "$o")
  if [ "$b" == "${SRC%.hip}.o" ]; then OBJS="$OBJS $OUT/$b"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libganq_hip.so" $OBJS
echo "$OUT/libganq_hip.so"
