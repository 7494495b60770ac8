#!/bin/bash
# Store-path experiments on conv3x3_wino2d_kernel ("a CU stores ~12 bytes per cycle"): build a diagnostic library (phase stamps + one experiment)
# HERE (no GPU needed), then run tools/w2_stamp.py on the GPU box; restore the product library afterwards with `python -m sleap_nn_amd.build --force`.
#   bash tools/w2_store_exp.sh build <exp>      exp: 0 base | 1 half the stores | 2 coalesced (wrong) addresses | 3 nt | 4 sc1 | 5 sc0 sc1 | 6 sc0 | s<pct> start stagger
#   bash tools/w2_store_exp.sh run              (on the GPU box)
cd "$(dirname "$0")/.."
case "$1" in
  build)
    if [[ "$2" == s* ]]; then FLAGS="-DPH_W2_STAMP -DPH_W2_STAGGER=${2#s}"; else FLAGS="-DPH_W2_STAMP -DPH_W2_STORE_EXP=${2:-0}"; fi
    PH_EXTRA_HIPCC_FLAGS="$FLAGS" python -m sleap_nn_amd.build --force | tail -1 ;;
  run) python tools/w2_stamp.py 2>/dev/null | head -10 ;;
  *) echo "usage: $0 build <exp> | run" ;;
esac
