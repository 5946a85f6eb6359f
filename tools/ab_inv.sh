#!/bin/bash
# A/B of library builds on the fused inverse: bash tools/ab_inv.sh libA.so libB.so ...
lib=mlx_audio_amd/csrc/libmia_hip.so
cp $lib /tmp/ab_saved.so
echo "== shipped"; timeout -k 10 120 python -u tools/time_istft_resident.py 2>/dev/null | grep n_fft || exit 1
for v in "$@"; do
  cp $v $lib
  echo "== $v"; timeout -k 10 120 python -u tools/time_istft_resident.py 2>/dev/null | grep n_fft || { cp /tmp/ab_saved.so $lib; exit 1; }
done
cp /tmp/ab_saved.so $lib
