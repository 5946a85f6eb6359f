#!/bin/bash
# Ablation timings of the Whisper kernel (library built with MIA_EXTRA_FLAGS="-DMIA_DEBUG_HOOKS"):
# MIA_DBG bit 1 skips the mel sweep, bit 4 the output stores.
for d in "$@"; do echo "dbg=$d"; MIA_DBG=$d timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1; done
