#!/bin/bash
# A/B of the mel projection (sweep vs matrix cores) with counters, one box, one binary.
# Writes gpurun_out/mfma_ab/report.txt
set -u
root=$(pwd)
out=$root/gpurun_out/mfma_ab
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/mfma_ab.py > "$out/timing.txt" 2>&1 || exit 1
cd /tmp
for v in compiled mfma dma; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d "$out/$v" -- \
    python3 "$root/tools/mfma_ab.py" $v > "$out/$v.log" 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d "$out/${v}_b" -- \
    python3 "$root/tools/mfma_ab.py" $v > "$out/${v}_b.log" 2>&1 || exit 1
done
cd "$root"
python3 - "$out" > "$out/report.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
print(open(out + "/timing.txt").read())
for v in ("compiled", "mfma", "dma"):
    agg = defaultdict(lambda: [0, 0.0])
    for sub in (v, v + "_b"):
        for path in sorted(glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)):
            for row in csv.DictReader(open(path)):
                if "logmel400" not in row["Kernel_Name"]:
                    continue
                agg[row["Counter_Name"]][0] += 1
                agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
    print(f"--- variant {v}: counters of the logmel400 kernel, mean per launch (1024 x 30 s; 4096 waves, 256 CUs)")
    for c, (n, val) in sorted(agg.items()):
        print(f"  {c:30s} launches={n:3d} mean={val / n:.6g}")
PY
cat "$out/report.txt"
