#!/bin/bash
# Builds the microbenchmarks next to their sources (run here, on the CPU; the binaries travel
# to the GPU box with the tree): bash tools/ubench/build.sh
set -e
cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o "${f%.hip}" "$f"
done
