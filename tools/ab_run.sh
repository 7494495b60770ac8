#!/bin/bash
# A/B of library builds on one box: bash tools/ab_run.sh <script args...>; runs the script once per tools/ab/lib_*.so (copied over the in-tree library)
cd $GRAFT_REPO_ROOT
for f in tools/ab/lib_*.so; do
  cp $f sleap_nn_amd/lib/libposehip.so
  echo "== $f"
  timeout -k 10 200 python "$@" 2>&1 | grep -v amdgpu.ids
done
