#!/bin/bash
# A/B of library builds on chosen workloads: tools/ab_w.sh "w1 w2" libA.so libB.so ...
lib=mlx_audio_amd/csrc/libmia_hip.so
ws="$1"; shift
cp $lib /tmp/ab_saved.so
for round in 1 2; do
  for f in "$@"; do
    cp "$f" $lib
    for w in $ws; do
      timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('round $round $f $w', round(d['roofline']['device_ms_per_step'],4), 'ms', round(d['roofline']['frac']*100,1), '%')" || exit 1
    done
  done
done
cp /tmp/ab_saved.so $lib
