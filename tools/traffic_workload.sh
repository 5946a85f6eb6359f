#!/bin/bash
# HBM traffic of one bench workload: FETCH_SIZE and WRITE_SIZE in separate passes, summed over
# the mia:: kernels of one step (KiB per step printed).  bash tools/traffic_workload.sh <workload>
set -u
w=${1:-clap64}
root=$(pwd); out=$root/gpurun_out/traffic_$w
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -- \
    python3 "$root/bench.py" --workload "$w" --steps 2 --warmup 1 --no-cpu-baseline > "$out/$c.log" 2>&1 || exit 1
done
cd "$root"
python3 - "$out" "$w" <<'PY'
# Per step = everything the library's kernels moved, divided by the number of steps, which is the
# dispatch count of a kernel launched once per step (the smallest count among the kernels).
import csv, glob, sys
from collections import defaultdict
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per_kernel = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(f"{sys.argv[1]}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "mia::" in row["Kernel_Name"] and row["Counter_Name"] == c:
                k = row["Kernel_Name"].split("(")[0][:60]
                per_kernel[k][0] += 1
                per_kernel[k][1] += float(row["Counter_Value"])
    steps = min(n for n, _ in per_kernel.values())
    for k, (n, v) in sorted(per_kernel.items()):
        print(f"{sys.argv[2]} {c} {k}: {v / n:.0f} KiB per dispatch, {n / steps:g} dispatches per step")
    res[c] = sum(v for _, v in per_kernel.values()) / steps
print(f"{sys.argv[2]} per step: fetch {res['FETCH_SIZE']:.0f} KiB (x2 on gfx950), write {res['WRITE_SIZE']:.0f} KiB, "
      f"hbm bytes {int((2 * res['FETCH_SIZE'] + res['WRITE_SIZE']) * 1024)}")
PY
