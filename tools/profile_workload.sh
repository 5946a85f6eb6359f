#!/bin/bash
# Kernel trace of one bench workload: bash tools/profile_workload.sh <workload> <tag>
set -u
w=${1:-clap64}; tag=${2:-x}
root=$(pwd); out=$root/gpurun_out/prof_${w}_$tag
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- \
  python3 "$root/bench.py" --workload "$w" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$out/bench.log" 2>&1 || exit 1
cd "$root"
for f in $(find "$out/trace" -name "*kernel_stats.csv"); do head -8 "$f" | cut -c1-150; done
