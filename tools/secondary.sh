#!/bin/bash
# Bench lines of the secondary workloads (generic kernels).
for w in clap64 mfcc40 htdemucs; do
  timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac']*100,1), '%')" || exit 1
done
