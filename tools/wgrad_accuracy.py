"""Weight-gradient accuracy of wgrad_wino = 1 / 0 against the float64 oracle (cfg3 at 256x320, B = 2, as the GPU test) and the
sensitivity of three Adam steps to it (the small net of test_adam_steps_match_torch_optim).  python tools/wgrad_accuracy.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import bench
from oracle import cpu_ref as O
from tests import test_gpu_training as T

bb, heads, mt = dict(bench.CFG3_BB), {k: dict(v) for k, v in bench.CFG3_HEADS.items()}, "bottomup"
for v in (1, 0):
    sd, img, targets, lw, tm = T._setup(bb, heads, mt, (256, 320), 2, seed=3)
    tm.model.set_option("wgrad_wino", v)
    if v == 1:
        _, g32 = O.training_step(sd, bb, heads, mt, img, targets, lw)
        _, g64 = O.training_step({k: x.double() for k, x in sd.items()}, bb, heads, mt, img, {k: x.double() for k, x in targets.items()}, lw)
    tm.forward_backward(img, targets)
    got = tm.named_grads()
    rows = []
    for k, r in g64.items():
        if not k.endswith(".weight"):
            continue
        scale = max(float(r.abs().max()), 1e-30)
        rows.append((float((got[k].double() - r).abs().max()) / scale, float((g32[k].double() - r).abs().max()) / scale, k))
    rows.sort(reverse=True)
    print(f"wgrad_wino {v}: worst weight-gradient errors vs float64 (hip, torch fp32, tensor):")
    for r in rows[:5]:
        print("   %.3g %.3g %s" % r)
    print("   median hip %.3g torch %.3g" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])), flush=True)

bb, heads, mt = T._cfg(8, 8, 2)
for v in (1, 0):
    sd, img, targets, lw, tm = T._setup(bb, heads, mt, (48, 64), 2, seed=3, lr=1e-3, amsgrad=False, optimizer="Adam")
    tm.model.set_option("wgrad_wino", v)
    grads_seq, cur = [], {k: x.clone() for k, x in sd.items()}
    for step in range(3):
        _, g = O.training_step(cur, bb, heads, mt, img, targets, lw)
        grads_seq.append(g)
        cur = O.adam_reference(sd, grads_seq, lr=1e-3, amsgrad=False, optimizer="Adam")
        tm.training_step({"image": img, **targets})
    got = tm.state_dict()
    dev = sorted(((float((got[k] - r).abs().max()), k) for k, r in cur.items()), reverse=True)
    print(f"adam x3, wgrad_wino {v}: worst parameter deviations", dev[:3], flush=True)
