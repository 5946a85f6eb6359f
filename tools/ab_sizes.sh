#!/bin/bash
# tools/ab_sizes.sh N_FFT lib1.so lib2.so ...: tools/ab_size.py under each library build
n=$1; shift
lib=mlx_audio_amd/csrc/libmia_hip.so
cp $lib /tmp/ab_saved.so
for f in "$@"; do
  cp "$f" $lib
  echo -n "$f  "; timeout -k 10 120 python tools/ab_size.py $n 2>&1 | grep -v amdgpu || exit 1
done
cp /tmp/ab_saved.so $lib
