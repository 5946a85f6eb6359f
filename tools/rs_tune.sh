#!/bin/bash
# Polyphase resampler: parity tests, then the bench workload over the launch-shape knobs.
set -e
cd "$(dirname "$0")/.."
timeout -k 10 300 python -u -m pytest tests/test_gpu_resample.py -x -q --timeout 120 --timeout-method thread > gpurun_out/rs_test.log 2>&1 || { tail -30 gpurun_out/rs_test.log; exit 1; }
tail -2 gpurun_out/rs_test.log
for thr in ${RS_THREADS:-320 512}; do
  for k in ${RS_K:-8}; do
    echo -n "threads~$thr k=$k: "
    MIA_RS_THREADS=$thr MIA_RS_K=$k timeout -k 10 120 python bench.py --workload resample --no-cpu-baseline --steps 20 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_step'],3),'ms', round(100*d['roofline']['frac'],1),'%')"
  done
done
