#!/bin/bash
# Ablation timing of the Whisper kernel (library built with -DMIA_DEBUG_HOOKS): MIA_DBG bits
# 1 = no mel projection, 4 = no output stores.
for d in 0 1 4; do
  echo "MIA_DBG=$d"
  MIA_DBG=$d timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1
done
