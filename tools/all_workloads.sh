#!/bin/bash
# One bench line per workload on one box: bash tools/all_workloads.sh > gpurun_out/all_workloads.txt
for w in whisper128 clap64 mfcc40 htdemucs resample resample_fft features features_all features_y contrast onset griffinlim spectral_gate demucs_front whisper_stem; do
  if [ "$w" = whisper128 ]; then args=""; else args="--workload $w"; fi
  timeout -k 10 300 python bench.py $args --no-cpu-baseline --no-secondary | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-14s %9.4f ms/step  %8.1f %s  %5.1f %% of %s peak  traffic/algorithmic %s' % ('$w', r.get('device_ms_per_step', d['ms_per_step']), d['value'], d['unit'], 100*r['frac'], r['bound'], ('%.2f' % (r['traffic']/r['algorithmic_bytes_per_step'])) if r.get('traffic') and r.get('algorithmic_bytes_per_step') else 'n/a'))" || exit 1
done
