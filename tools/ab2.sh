#!/bin/bash
# A/B of library builds on the secondary workloads: tools/ab2.sh libA.so libB.so ...
lib=mlx_audio_amd/csrc/libmia_hip.so
cp $lib /tmp/ab_saved.so
for round in 1 2; do
  for f in "$@"; do
    cp "$f" $lib
    echo "round $round $f"
    tools/secondary.sh || exit 1
  done
done
cp /tmp/ab_saved.so $lib
