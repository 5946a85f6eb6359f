#!/bin/bash
# Any timing script under several library builds: bash tools/ab_lib.sh "python tools/x.py args" libA.so libB.so ...
lib=mlx_audio_amd/csrc/libmia_hip.so
cmd="$1"; shift
cp $lib /tmp/ab_saved.so
for f in "$@"; do
  cp "$f" $lib
  echo "== $f"; timeout -k 10 200 $cmd 2>/dev/null || { cp /tmp/ab_saved.so $lib; exit 1; }
done
cp /tmp/ab_saved.so $lib
