#!/bin/bash
# LDS bank-conflict share of every transform kernel: bash tools/lds_scan.sh > gpurun_out/lds_scan.txt
root=$(pwd); out=$root/gpurun_out/lds_scan; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
timeout -s KILL 170 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d "$out/a" -- \
  python3 "$root/tools/lds_scan.py" > "$out/a.log" 2>&1 || exit 1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "mia::" in row["Kernel_Name"]:
            agg[row["Kernel_Name"].split("(")[0][:110]][row["Counter_Name"]] += float(row["Counter_Value"])
rows = []
for k, c in agg.items():
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    rows.append((c.get("SQ_LDS_BANK_CONFLICT", 0.0) / act if act else 0.0,
                 c.get("SQ_WAIT_INST_LDS", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), k))
for conf, wait, k in sorted(rows, reverse=True):
    print(f"conflicts {100 * conf:5.1f} % of LDS cycles   waiting on LDS {100 * wait:5.1f} % of wave cycles   {k}")
PY
