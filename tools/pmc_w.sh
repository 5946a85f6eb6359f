#!/bin/bash
# SQ counter passes over one bench workload (each pass its own run).
# Usage: bash tools/pmc_w.sh <workload> <kernel-name substring> <tag>
set -u
w=${1:-clap64}
kern=${2:-fwd_kernel}
tag=${3:-x}
root=$(pwd)
out=$root/gpurun_out/pmcw_${w}_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/a" -- \
  python3 "$root/bench.py" --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/a.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/b" -- \
  python3 "$root/bench.py" --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/b.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/c" -- \
  python3 "$root/bench.py" --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/c.log" 2>&1 || exit 1
cd "$root"
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(path)):
        if sys.argv[2] not in row["Kernel_Name"]:
            continue
        agg[row["Counter_Name"]][0] += 1
        agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
for c, (n, v) in sorted(agg.items()):
    print(f"{c:26s} n={n} mean={v / n:.6g}")
PY
