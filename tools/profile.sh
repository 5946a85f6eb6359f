#!/bin/bash
# Round profile: kernel trace + HBM traffic counters of the default bench command.
# Usage (on the GPU box, from the repo root): bash tools/profile.sh <tag>
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- \
  python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_trace.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- \
  python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$out/bench_write.log" 2>&1 || exit 1
cd "$root"
python3 tools/summarize_profile.py "$out" > "$out/summary.txt" 2>&1
tail -30 "$out/summary.txt"
