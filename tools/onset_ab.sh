#!/bin/bash
# A/B of the onset flux kernel's logarithm: swaps the library in place on the GPU box.
set -e
cd "$(dirname "$0")/.."
L=mlx_audio_amd/csrc
cp $L/libmia_hip.so /tmp/base.so
for v in hip logffx __logffx; do
  if [ $v = hip ]; then cp /tmp/base.so $L/libmia_hip.so; else cp $L/libmia_$v.so $L/libmia_hip.so; fi
  echo "== $v"
  timeout -k 10 200 python -u -m pytest tests/test_gpu_onset.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1
  timeout -k 10 100 python bench.py --workload onset --steps 30 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_step'],3),'ms', round(100*d['roofline']['frac'],1),'%')"
done
cp /tmp/base.so $L/libmia_hip.so
