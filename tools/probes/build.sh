#!/bin/bash
# Build the stand-alone hardware probes next to their sources (run on the GPU box): bash tools/probes/build.sh
cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o "${f%.hip}" "$f" || exit 1
done
