"""Per-op timing of the ConvNeXt-tiny centered-instance network (cfg4 shapes: 384x384 crops, os=2).

    python tools/convnext_bench.py [batch] [size]
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from sleap_nn_amd.architectures.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 384
bb = {"model_type": "tiny", "arch": None, "in_channels": 1, "kernel_size": 3, "filters_rate": 2, "convs_per_block": 2, "up_interpolate": True,
      "stem_patch_kernel": 4, "stem_patch_stride": 2, "output_stride": 2, "max_stride": 32}
heads = {"confmaps": {"part_names": [str(i) for i in range(13)], "sigma": 2.5, "output_stride": 2}}
t0 = time.time()
m = Model("convnext", bb, heads, "centered_instance")
m.init_xavier_(seed=1234, head_scale=0.05)
m.to("cuda:0")
img = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, device="cuda:0")
m(img)
torch.cuda.synchronize()
print("setup s", time.time() - t0)
for _ in range(2):
    m(img)
torch.cuda.synchronize()
t = time.time()
N = 5
for _ in range(N):
    m(img)
torch.cuda.synchronize()
dt = (time.time() - t) / N
rows = m.op_table(B, S, S)
tot = sum(r["flops"] for r in rows)
print(f"forward {dt*1e3:.2f} ms  {B/dt:.1f} crops/s  {tot/dt/1e12:.1f} TFLOP/s")
m.set_profiling(True)
for _ in range(N):
    m(img)
ms, n = m.read_profile()
agg = {}
for r, t_ in zip(rows, ms):
    k = {2: "conv3x3", 3: "pool", 4: "upsample", 6: "head", 8: "patch_stem", 9: "dwconv", 10: "layernorm", 11: "linear", 12: "patch_conv"}.get(r["kind"], str(r["kind"]))
    a = agg.setdefault(k, [0.0, 0.0, 0.0, 0])
    a[0] += t_ / n
    a[1] += r["flops"]
    a[2] += r["bytes"]
    a[3] += 1
for k, (t_, f, by, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:12s} n={c:3d} {t_:8.3f} ms  {f/t_/1e9 if t_ else 0:8.1f} TFLOP/s  {by/t_/1e6 if t_ else 0:8.1f} GB/s")
det = [(r["label"], t_ / n, r["flops"] / (t_ / n) / 1e9 if t_ else 0) for r, t_ in zip(rows, ms)]
for lab, t_, tf in det[:14] + det[-22:]:
    print(f"  {lab[-44:]:44s} {t_:7.3f} ms {tf:7.1f} TF")
