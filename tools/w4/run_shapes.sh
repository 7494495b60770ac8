#!/bin/bash
# On the GPU box: the kernel on cfg3's F(4x4,3x3) layer shapes (32 frames) and two ragged ones
cd "$(dirname "$0")/../.."
f=${1:-tools/w4/w4_bench_0}
for s in "32 64 64 256 0 256 0" "32 64 64 256 512 256 1" "32 128 128 128 256 128 1" "32 256 256 64 128 64 1" "32 128 128 128 0 128 0" "32 256 256 64 0 64 0" "32 32 32 256 0 512 0" "32 32 32 512 0 512 0" "3 40 72 64 64 96 0" "2 24 40 64 64 96 1"; do
  timeout -k 5 60 $f $s 20 || exit 1
done
