"""Error of every layer of the cfg3 network against a float64 evaluation of the same network, per arithmetic:

    exact   fp32 products on the fp32 matrix pipe (Winograd F(4x4,3x3) / F(2x2,3x3) kernels: the benched path, `dtype f32`)
    split   operands as (hi, lo') fp16 pairs, three fp16 MFMAs per product, fp32 accumulation (direct kernels on the 16-bit pipe)
    fp16    the reference's autocast mode

VERDICT r4 item 8: an emulation on the 16-bit pipe may only carry the headline if its error against the fp64 convolution is <= the exact-fp32 kernel's on
EVERY cfg3 layer.  This prints that table (max |y - y64| / max |y64| per layer output, accumulated through the network: what a user of the layer sees) and the verdict.

    python tools/precision_table.py [frames=1] [size=1024]
"""
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from oracle import cpu_ref as O
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05)
sd = {k: v.clone() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(4321)
img = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, generator=g)
t0 = time.perf_counter()
collect = {}
with torch.inference_mode():
    ref = O.model_forward({k: v.double() for k, v in sd.items()}, bench.CFG3_BB, bench.CFG3_HEADS, "bottomup", img, collect=collect)
print(f"# float64 network on the host: {time.perf_counter() - t0:.1f} s for {B} frame(s) of {S} x {S}", flush=True)
m = m.to(dev)
rows = {}
kernels = {}
for prec in ("exact", "split", "fp16"):
    m.set_precision(prec).set_keep_activations(True)
    out = m(img.to(dev))
    torch.cuda.synchronize()
    if prec == "exact":
        codes = m.last_kernels()
        for r, c in zip(m.op_table(B, S, S), codes):
            kernels[r["label"].split("+")[0]] = L.KV_NAMES.get(c, "-").split(" ")[0]
    for lab, t in collect.items():
        if lab not in m.backbone.labels:
            continue
        try:
            got = m.read_activation(lab, t.shape[0], t.shape[-2:]).cpu().double()
        except KeyError:
            continue  # fused away (the first conv of the stem never exists in HBM)
        rows.setdefault(lab, {})[prec] = float((got - t).abs().max() / t.abs().max())
    for k, v in ref.items():
        rows.setdefault("head: " + k, {})[prec] = float((out[k].cpu().double() - v).abs().max() / v.abs().max())
print(f"{'layer output':46s} {'exact-path kernel':24s} {'exact fp32':>11s} {'split f16x3':>12s} {'fp16':>10s}   split <= exact")
worse = []
for lab, e in rows.items():
    ok = e["split"] <= e["exact"]
    if not ok:
        worse.append(lab)
    print(f"{lab:46s} {kernels.get(lab, ''):24s} {e['exact']:11.2e} {e['split']:12.2e} {e['fp16']:10.2e}   {'yes' if ok else 'NO (x%.1f)' % (e['split'] / e['exact'])}")
print(f"# split-fp16 is at or below the exact-fp32 kernels' error on {len(rows) - len(worse)} of {len(rows)} layer outputs"
      + ("" if not worse else f"; above it on: {', '.join(worse)}"))
