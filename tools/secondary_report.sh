#!/bin/bash
# Bench line, kernel trace, HBM traffic and SQ counters of the three BASELINE workloads that run
# on the generic kernels; one box.  Writes gpurun_out/secondary_report.txt
set -u
root=$(pwd)
rep=$root/gpurun_out/secondary_report.txt
: > "$rep"
for w in clap64 mfcc40 htdemucs; do
  echo "== $w" >> "$rep"
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline >> "$rep" 2>/dev/null || exit 1
  bash tools/profile_workload.sh $w r03 2>/dev/null | grep "mia::" | cut -c1-200 >> "$rep" || exit 1
  bash tools/traffic_workload.sh $w 2>/dev/null | grep -v amdgpu.ids >> "$rep" || exit 1
done
for spec in "clap64 fwd_kernel" "mfcc40 fwd_kernel" "htdemucs fwd_kernel" "htdemucs istft_own_kernel"; do
  set -- $spec
  echo "== SQ counters: $1, kernels matching $2 (mean per launch; *_CYCLES / ACTIVE / WAIT tick once per 4 clocks)" >> "$rep"
  bash tools/pmc_w.sh $1 $2 r03_$2 2>/dev/null | grep -v amdgpu.ids >> "$rep" || exit 1
done
cat "$rep"
