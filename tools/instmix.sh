#!/bin/bash
# Instruction mix of the Whisper kernel per wave and tile (one PMC pass per MIA_DBG setting;
# settings other than 0 need a library built with -DMIA_DEBUG_HOOKS).
# Usage: bash tools/instmix.sh <tag> [dbg ...]
set -u
tag=${1:-x}
shift
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
for dbg in "${@:-0}"; do
  out=$root/gpurun_out/instmix_${tag}_$dbg
  mkdir -p "$out"
  export MIA_DBG=$dbg
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d "$out" -- \
    python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/run.log" 2>&1 || exit 1
  echo "MIA_DBG=$dbg"
  python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if "logmel400" not in row["Kernel_Name"]:
            continue
        agg[row["Counter_Name"]][0] += 1
        agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
    for c, (n, v) in sorted(agg.items()):
        print(f"  {c:22s} per wave and tile {v / n / 4096 / 188:8.1f}")
PY
done
