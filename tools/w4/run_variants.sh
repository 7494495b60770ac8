#!/bin/bash
# On the GPU box: run every built variant on the three decoder shapes of cfg3, two rounds interleaved
cd "$(dirname "$0")/../.."
for round in 1 2; do
  for f in tools/w4/w4_bench_*; do
    [ -x "$f" ] || continue
    echo "== $f"
    timeout -k 5 60 $f 32 64 64 256 512 256 1 20 | grep -v determinism
    timeout -k 5 60 $f 32 128 128 128 256 128 1 20 | grep -v determinism
    timeout -k 5 60 $f 32 256 256 64 128 64 1 20 | grep -v determinism
  done
done
