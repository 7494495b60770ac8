#!/bin/bash
# On the GPU box: every built ablation variant of one kernel (arg 1: 1 twelve-wave, 2 pipelined) on one plain and one folded-bilinear shape
cd "$(dirname "$0")/../.."
which=${1:-2}
for f in tools/w4/w4_bench_*; do
  [ -x "$f" ] || continue
  echo "== $f"
  timeout -k 5 60 $f 32 64 64 256 0 256 0 20 $which | grep -v "vs twelve" || exit 1
done
