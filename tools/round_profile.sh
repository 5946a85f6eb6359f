#!/bin/bash
# Everything the round's profile files are made from, on ONE box: the default bench line, the
# kernel trace and HBM traffic passes of the same command (tools/profile.sh), the SQ counter
# passes (tools/pmc.sh, tools/pmc2.sh), the Whisper stem line and trace, and the report of the
# BASELINE workloads that run on the generic kernels (tools/secondary_report.sh).
# Usage: bash tools/round_profile.sh <tag>     (writes gpurun_out/round_<tag>/)
set -u
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/round_$tag
mkdir -p "$out"
timeout -k 10 300 python3 bench.py > "$out/bench_line.json" 2> "$out/bench_line.err" || exit 1
bash tools/profile.sh $tag > "$out/profile.log" 2>&1 || exit 1
cp gpurun_out/prof_$tag/summary.txt "$out/bench_summary.txt"
cp $(find gpurun_out/prof_$tag/trace -name "*kernel_stats.csv" | head -1) "$out/bench_kernel_stats.csv"
bash tools/pmc.sh $tag > "$out/pmc.txt" 2>&1 || exit 1
bash tools/pmc2.sh $tag > "$out/pmc2.txt" 2>&1 || exit 1
timeout -k 10 300 python3 bench.py --workload whisper_stem --no-cpu-baseline > "$out/stem_line.json" 2>/dev/null || exit 1
bash tools/profile_workload.sh whisper_stem $tag 2>/dev/null | grep "mia" | cut -c1-220 > "$out/stem_kernels.txt" || exit 1
timeout -k 10 300 python3 bench.py --workload whisper_stem_f16 --no-cpu-baseline > "$out/stem_f16_line.json" 2>/dev/null || exit 1
bash tools/profile_workload.sh whisper_stem_f16 $tag 2>/dev/null | grep "mia" | cut -c1-220 > "$out/stem_f16_kernels.txt" || exit 1
timeout -k 10 120 python3 tools/w400_paths.py > "$out/w400_paths.txt" 2>/dev/null || exit 1
bash tools/secondary_report.sh > "$out/secondary.log" 2>&1 || exit 1
cp gpurun_out/secondary_report.txt "$out/secondary_report.txt"
echo "round profile done"; tail -c 600 "$out/bench_line.json"
