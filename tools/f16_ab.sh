#!/bin/bash
# per-op fp16 forward times (cfg5 shape) for each library build under tools/ab/ (bash tools/f16_ab.sh [pattern], GPU box, repo root)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
PAT=${1:-"enc0|enc1_conv0|forward"}
cp sleap_nn_amd/lib/libposehip.so /tmp/base.so
for f in tools/ab/lib_*.so; do
  cp $f sleap_nn_amd/lib/libposehip.so
  echo "== $f"
  timeout -k 10 200 python3 tools/f16_rows_check.py 16 768 only1 1 2>&1 | grep -E "$PAT"
done
cp /tmp/base.so sleap_nn_amd/lib/libposehip.so
