#!/bin/bash
# tools/build_variant.sh OUT.so FILE.hip "-DFLAG ..." : the library with one source rebuilt under extra flags
set -e
out=$1; src=$2; flags=$3
d=mlx_audio_amd/csrc
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function $flags -c $d/$src -o $tmp/v.o
objs=""
for f in $d/*.o; do
  if [ "$(basename $f .o)" != "$(basename $src .hip)" ]; then objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs $tmp/v.o
rm -rf $tmp
