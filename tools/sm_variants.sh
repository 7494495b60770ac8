#!/bin/bash
# Build timing variants of conv3x3_sm_kernel (ablation macros in smallmap_kernels.hip) into sleap_nn_amd/lib/variants/sm_<name>.so (here, no GPU needed);
# on the GPU box `bash tools/sm_variants.sh run <names...>` times BASELINE cfg1 with each (tools/small_batch.py).  Timing experiments only: most variants compute garbage.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = run ]; then
  shift
  cp sleap_nn_amd/lib/libposehip.so /tmp/base.so
  for v in base "$@"; do
    if [ "$v" = base ]; then cp /tmp/base.so sleap_nn_amd/lib/libposehip.so; else cp sleap_nn_amd/lib/variants/sm_$v.so sleap_nn_amd/lib/libposehip.so; fi
    echo "== $v"
    python tools/small_batch.py ${SM_CFG:-cfg1} 2>&1 | grep -E "sm_kernel|graph=" | cut -c1-44,68-140
  done
  cp /tmp/base.so sleap_nn_amd/lib/libposehip.so
  exit 0
fi
mkdir -p sleap_nn_amd/lib/variants
objs=$(ls sleap_nn_amd/lib/*.o | grep -v smallmap_kernels.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c sleap_nn_amd/csrc/smallmap_kernels.hip -o /tmp/sm_$name.o -I include $flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sleap_nn_amd/lib/variants/sm_$name.so $objs /tmp/sm_$name.o
  echo built sm_$name
done
