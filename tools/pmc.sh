#!/bin/bash
# SQ counter passes over the default bench (each pass its own run).
# Usage: bash tools/pmc.sh <tag>
set -u
tag=${1:-x}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/a" -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/a.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/b" -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/b.log" 2>&1 || exit 1
cd "$root"
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
        print(f"{c:26s} n={n} mean={v / n:.6g} per_wave={v / n / 4096:.6g}")
PY
