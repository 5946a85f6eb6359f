#!/bin/bash
# tools/ab_any.sh lib1.so lib2.so ...: tools/time_any_sizes.py under each library build
lib=mlx_audio_amd/csrc/libmia_hip.so
cp $lib /tmp/ab_saved.so
for f in "$@"; do
  cp "$f" $lib
  echo "== $f"; timeout -k 10 200 python tools/time_any_sizes.py 2>&1 | grep -v amdgpu || exit 1
done
cp /tmp/ab_saved.so $lib
