#!/bin/bash
# A/B of the radix-16 pass of the n_fft = 400 kernel: exchange through LDS + register
# butterflies (shipped) against the cross-lane form with DPP operands, one box, one binary,
# with the SQ counters that tell the two apart.  Writes gpurun_out/dpp_ab/report.txt
set -u
root=$(pwd)
out=$root/gpurun_out/dpp_ab
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/mfma_ab.py > "$out/timing.txt" 2>&1 || exit 1
cd /tmp
for v in dma dpp; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d "$out/$v" -- \
    python3 "$root/tools/mfma_ab.py" $v > "$out/$v.log" 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES --output-format csv -d "$out/${v}_b" -- \
    python3 "$root/tools/mfma_ab.py" $v > "$out/${v}_b.log" 2>&1 || exit 1
done
cd "$root"
python3 - "$out" > "$out/report.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
print(open(out + "/timing.txt").read())
for v in ("dma", "dpp"):
    agg = defaultdict(lambda: [0, 0.0])
    for sub in (v, v + "_b"):
        for path in sorted(glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)):
            for row in csv.DictReader(open(path)):
                if "logmel400" not in row["Kernel_Name"]:
                    continue
                agg[row["Counter_Name"]][0] += 1
                agg[row["Counter_Name"]][1] += float(row["Counter_Value"])
    print(f"--- variant {v}: counters of the logmel400d kernel, mean per launch (1024 x 30 s; 4096 waves, 770048 wave-tiles)")
    for c, (n, val) in sorted(agg.items()):
        print(f"  {c:30s} launches={n:3d} mean={val / n:.6g}  per wave-tile={val / n / 770048:.1f}")
PY
cat "$out/report.txt"
