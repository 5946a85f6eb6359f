#!/bin/bash
# Ablation timings of the Whisper kernel (library built with MIA_EXTRA_FLAGS="-DMIA_DEBUG_HOOKS"):
# MIA_DBG bit 1 skips the mel sweep, 2 the FFT arithmetic, 4 the output stores, 16 the global loads.
# usage: tools/abl.sh 0 1 4 ...   (results are wrong by construction; only the time matters)
for d in "$@"; do echo -n "dbg=$d  ms_per_step="; MIA_DBG=$d timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1; done
