#!/bin/bash
# A/B timing of library builds on one box: tools/ab.sh libA.so libB.so [...]; each is copied
# over mlx_audio_amd/csrc/libmia_hip.so in turn, three rounds, bench.py ms_per_step printed.
set -u
lib=mlx_audio_amd/csrc/libmia_hip.so
cp $lib /tmp/ab_saved.so
for round in 1 2 3; do
  for f in "$@"; do
    cp "$f" $lib
    ms=$(timeout -k 10 120 python bench.py --steps 40 --warmup 8 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])") || exit 1
    echo "round $round $f $ms"
  done
done
cp /tmp/ab_saved.so $lib
