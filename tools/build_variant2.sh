#!/bin/bash
# tools/build_variant2.sh OUT.so "-DFLAG ..." FILE.hip [FILE.hip ...]: the library with several sources rebuilt under extra flags
set -e
out=$1; flags=$2; shift 2
d=mlx_audio_amd/csrc
tmp=$(mktemp -d)
skip=""
for src in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function $flags -c $d/$src -o $tmp/$(basename $src .hip).o &
  skip="$skip $(basename $src .hip)"
done
wait
objs=""
for f in $d/*.o; do
  b=$(basename $f .o)
  if ! echo " $skip " | grep -q " $b "; then objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs $tmp/*.o
rm -rf $tmp
