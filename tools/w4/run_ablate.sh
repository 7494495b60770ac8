#!/bin/bash
# On the GPU box: every built ablation variant (tools/w4/build_variants.sh "0 2 8 ...") on one plain layer shape
cd "$(dirname "$0")/../.."
for f in tools/w4/w4_bench_*; do
  [ -x "$f" ] || continue
  echo "== $f"
  timeout -k 5 60 $f 32 64 64 256 0 256 0 20 || exit 1
done
