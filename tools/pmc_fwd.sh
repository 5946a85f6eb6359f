#!/bin/bash
# SQ counters of the generic forward kernel on one bench workload: bash tools/pmc_fwd.sh <workload>
set -u
w=${1:-mfcc40}
root=$(pwd); out=$root/gpurun_out/pmc_fwd_$w
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d "$out/a" -- \
  python3 "$root/bench.py" --workload $w --steps 2 --warmup 1 --no-cpu-baseline > "$out/a.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d "$out/b" -- \
  python3 "$root/bench.py" --workload $w --steps 2 --warmup 1 --no-cpu-baseline > "$out/b.log" 2>&1 || exit 1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if "fwd_kernel" not in row["Kernel_Name"]:
            continue
        agg[row["Counter_Name"]][0] += 1
        agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
    for c, (n, v) in sorted(agg.items()):
        print(f"{c:26s} n={n} per dispatch {v / n:.6g}")
PY
