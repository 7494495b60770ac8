#!/bin/bash
# Time the bench's forward with alternative builds of the native library (sleap_nn_amd/lib/variants/*.so): timing experiments only.
cp sleap_nn_amd/lib/libposehip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so sleap_nn_amd/lib/libposehip.so; else cp sleap_nn_amd/lib/variants/$v.so sleap_nn_amd/lib/libposehip.so; fi
  python bench.py --no-cpu-baseline --no-extra-legs --steps 10 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; p=r['per_op_ms']; print('$v', round(d['value'],1), 'fwd', round(r['forward_ms'],3), 'conv', round(r['kernel_ms_per_forward'],3), 'dec0', p['stack0_dec0_s32_to_s16_refine_conv0'], 'enc2c1', p['stack0_enc2_conv1+pool'], 'enc1', p['stack0_enc1_conv0'], p['stack0_enc1_conv1+pool'])"
done
cp /tmp/base.so sleap_nn_amd/lib/libposehip.so
