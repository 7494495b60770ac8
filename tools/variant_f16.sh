#!/bin/bash
# build + time variants of f16_kernels.hip given as -D flags:  bash tools/variant_f16.sh "-DPH_EXP=1" "-DPH_EXP=2" ...
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  PH_EXTRA_HIPCC_FLAGS="$V" python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from sleap_nn_amd import build as b
os.utime(os.path.join(b.CSRC, "f16_kernels.hip"))
b.build()
PY
  echo "=== variant [$V]"
  python3 tools/f16_speed.py split 32 2>&1 | tail -2
done
