#!/bin/bash
# Issue / LDS / wait counters of the per-wave transform at one framing (each pass its own run).
set -u
root=$(pwd)
out=$root/gpurun_out/pmc_wave
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d "$out/a" -- \
  python3 "$root/tools/run_stft_once.py" "$@" > "$out/a.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/b" -- \
  python3 "$root/tools/run_stft_once.py" "$@" > "$out/b.log" 2>&1 || exit 1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if "wave_kernel" not in row["Kernel_Name"]:
            continue
        agg[row["Counter_Name"]][0] += 1
        agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
    for c, (n, v) in sorted(agg.items()):
        print(f"{c:26s} n={n} mean={v / n:.6g}")
PY
