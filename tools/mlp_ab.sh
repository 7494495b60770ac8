#!/bin/bash
# stage-0 CNBlock MLP time for each library build under tools/ab/ (bash tools/mlp_ab.sh, GPU box, repo root)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
cp sleap_nn_amd/lib/libposehip.so /tmp/base.so
for f in tools/ab/lib_*.so; do
  cp $f sleap_nn_amd/lib/libposehip.so
  echo "== $f"
  timeout -k 10 200 python3 - 2>&1 <<PY | grep -E "^forward|^linear|^  3 " | head -5
import sys
sys.argv=["x","64","384"]
src=open("tools/convnext_bench.py").read().replace("det[:14] + det[-22:]","det[:16]")
exec(compile(src,"cb","exec"))
PY
done
cp /tmp/base.so sleap_nn_amd/lib/libposehip.so
