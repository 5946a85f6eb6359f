#!/bin/bash
# Copies what tools/round_profile.sh <tag> left under gpurun_out/round_<tag>/ into the tracked
# profiles/<tag>_* files (run here, after the GPU call has merged gpurun_out/ back).
set -eu
tag=${1:-r03}
src=gpurun_out/round_$tag
cp $src/bench_line.json profiles/${tag}_bench_line.json
cp $src/bench_kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
cp $src/bench_summary.txt profiles/${tag}_bench_summary.txt
{
  echo "SQ counters of logmel400d_kernel<1, 128, 0> on the default bench (1024 x 30 s; mean per launch; per_wave = / 4096 waves; *_CYCLES, ACTIVE_* and WAIT_* tick once per 4 clocks).  tools/pmc.sh, tools/pmc2.sh, one pass per block."
  cat $src/pmc.txt
  echo
  cat $src/pmc2.txt
} > profiles/${tag}_sq_counters.txt
{
  echo "Whisper encoder stem, 64 spectrograms (128 x 3000) -> (64, 1500, 1280): bench lines and kernel trace per precision."
  echo "== float32 operands on v_mfma_f32_32x32x2_f32 (default for float32 weights)"
  cat $src/stem_line.json $src/stem_kernels.txt
  echo "== float16 operands, float32 accumulation (on request or for float16 weights)"
  cat $src/stem_f16_line.json $src/stem_f16_kernels.txt
} > profiles/${tag}_whisper_stem.txt
cp $src/secondary_report.txt profiles/${tag}_secondary_workloads.txt
cp $src/w400_paths.txt profiles/${tag}_w400_paths.txt
echo "collected into profiles/${tag}_*"
