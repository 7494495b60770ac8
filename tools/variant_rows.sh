#!/bin/bash
# build + time variants of f16_rows_kernels.hip given as -D flags:  bash tools/variant_rows.sh "-DPH_ROWS_EXP=1" ...   (GPU box, repo root; few instantiations)
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  PH_EXTRA_HIPCC_FLAGS="-DPH_ROWS_FEW $V" python3 - <<PY
import os, sys
sys.path.insert(0, ".")
from sleap_nn_amd import build as b
os.utime(os.path.join(b.CSRC, "f16_rows_kernels.hip"))
b.build()
PY
  echo "=== variant [$V]"
  python3 tools/f16_rows_check.py 16 768 only2 2>&1 | grep -v "^mode\|amdgpu.ids"
done
