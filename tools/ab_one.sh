#!/bin/bash
# A/B of library builds on ONE bench workload: tools/ab_one.sh <workload> libA.so libB.so [...]
set -u
w=$1; shift
lib=mlx_audio_amd/csrc/libmia_hip.so
cp $lib /tmp/ab_saved.so
for round in 1 2; do
  for f in "$@"; do
    cp "$f" $lib
    ms=$(timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])") || exit 1
    echo "round $round $f $w $ms"
  done
done
cp /tmp/ab_saved.so $lib
