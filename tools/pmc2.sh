#!/bin/bash
# LDS / VALU / busy counter passes over the default bench (each pass its own run).
set -u
tag=${1:-x}
root=$(pwd)
out=$root/gpurun_out/pmc2_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$out/a" -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/a.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$out/b" -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/b.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$out/c" -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/c.log" 2>&1 || exit 1
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
        print(f"{c:26s} n={n} mean={v / n:.6g} per_CU(256)={v / n / 256:.6g}")
PY
