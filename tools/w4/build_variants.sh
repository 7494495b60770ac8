#!/bin/bash
# Build the wino4 timing harness in several experiment variants (here, no GPU needed): bash tools/w4/build_variants.sh "0 1 2 4 8 16 3 ..."
cd "$(dirname "$0")/../.."
for v in ${1:-0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DW4_EXP=$v ${W4_FLAGS} -o tools/w4/w4_bench_$v tools/w4/w4_bench.hip 2>/dev/null || echo "build $v failed"
done
ls tools/w4
