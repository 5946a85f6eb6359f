#!/bin/bash
# A/B of library builds on the Whisper stem: parity tests, then the bench line, per build.
lib=mlx_audio_amd/csrc/libmia_hip.so
cp $lib /tmp/ab_saved.so
for f in "$@"; do
  cp "$f" $lib
  timeout -k 10 300 python -u -m pytest tests/test_gpu_whisper_stem.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1
  for r in 1 2; do
    timeout -k 10 200 python bench.py --workload whisper_stem --steps 10 --warmup 3 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', round(d['roofline']['device_ms_per_step'],4), 'ms', round(d['roofline']['achieved'],1), 'TFLOP/s')" || exit 1
  done
done
cp /tmp/ab_saved.so $lib
