"""Randomised forward parity sweep (UNet + ConvNeXt configs with awkward channel counts / sizes) vs the oracle.

    python tools/stress_forward.py [n_cases] [seed] [option=value ...]     (handle options, e.g. conv_smallmap=2 forces conv3x3_sm_kernel onto every 3x3 conv it takes)
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import cpu_ref as O
from sleap_nn_amd.architectures.model import Model

pos = [a for a in sys.argv[1:] if "=" not in a]
opts = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
n_cases = int(pos[0]) if len(pos) > 0 else 30
rng = np.random.default_rng(int(pos[1]) if len(pos) > 1 else 0)
kinds_seen = {}
worst = 0.0
for case in range(n_cases):
    if case % 3 != 2:
        down = int(rng.integers(2, 5))
        os_ = int(2 ** rng.integers(0, min(3, down)))
        bb = {"in_channels": int(rng.choice([1, 3])), "kernel_size": 3, "filters": int(rng.choice([4, 6, 8, 12, 16, 20, 24, 40])),
              "filters_rate": float(rng.choice([1.0, 1.5, 2.0])), "max_stride": 2**down, "stem_stride": None, "middle_block": bool(rng.random() < 0.8),
              "up_interpolate": bool(rng.random() < 0.7), "stacks": 1, "convs_per_block": int(rng.integers(1, 4)), "output_stride": os_}
        mt = str(rng.choice(["single_instance", "bottomup", "multi_class_bottomup"]))
        names = [f"n{i}" for i in range(int(rng.integers(1, 7)))]
        heads = {"confmaps": {"part_names": names, "output_stride": os_}}
        if mt == "bottomup":
            if len(names) < 2:
                names.append("extra")
                heads["confmaps"]["part_names"] = names
            heads["pafs"] = {"edges": [[names[i], names[i + 1]] for i in range(len(names) - 1)], "output_stride": min(2**down, os_ * 2)}
        if mt == "multi_class_bottomup":
            heads["class_maps"] = {"classes": ["a", "b", "c"], "output_stride": os_}
        kind = "unet"
        sd = O.init_state(bb, heads, mt, seed=int(rng.integers(1 << 30)), head_scale=1.0)
        mult = 2**down
    else:
        ch0 = int(rng.choice([8, 16, 24, 40]))
        ss = int(rng.choice([2, 4]))
        os_ = int(rng.choice([1, 2, 4]))
        bb = {"model_type": None, "arch": {"depths": [int(rng.integers(1, 3)) for _ in range(4)], "channels": [ch0, ch0 * 2 - 8, ch0 * 3, ch0 * 4 + 8]}, "in_channels": int(rng.choice([1, 3])),
              "kernel_size": 3, "filters_rate": float(rng.choice([1.5, 2.0])), "convs_per_block": int(rng.integers(1, 4)), "up_interpolate": bool(rng.random() < 0.7),
              "stem_patch_kernel": int(rng.choice([2, 4, 7])) if ss == 2 else 4, "stem_patch_stride": ss, "output_stride": os_, "max_stride": 32}
        mt = "single_instance"
        heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": os_}}
        kind = "convnext"
        sd = O.init_state_convnext(bb, heads, mt, seed=int(rng.integers(1 << 30)), head_scale=1.0, layer_scale=0.4, randomize_affine=True)
        mult = ss * 16
    B = int(rng.integers(1, 4))
    H, W = mult * int(rng.integers(1, 5)), mult * int(rng.integers(1, 5))
    g = torch.Generator().manual_seed(case)
    img = torch.randint(0, 256, (B, bb["in_channels"], H, W), dtype=torch.uint8, generator=g)
    try:
        ref = O.model_forward(sd, bb, heads, mt, img, backbone=kind)
        m = Model(kind, bb, heads, mt)
        m.load_state_dict(sd)
        for k_, v_ in opts.items():
            m.set_option(k_, v_)
        m.to("cuda:0")
        out = m(img.to("cuda:0"))
        torch.cuda.synchronize()
        for c_ in m.last_kernels():
            kinds_seen[c_] = kinds_seen.get(c_, 0) + 1
    except Exception as e:  # configs the product rejects on purpose must be rejected by a clear error
        print(f"case {case}: {kind} {type(e).__name__}: {str(e)[:120]}")
        continue
    err = max(float((out[k].cpu() - v).abs().max()) / max(1.0, float(v.abs().max())) for k, v in ref.items())
    worst = max(worst, err)
    flag = "" if err <= 1e-4 else "   <-- FAIL"
    print(f"case {case}: {kind} {mt} B={B} {H}x{W} bb={ {k: v for k, v in bb.items() if k in ('filters', 'filters_rate', 'max_stride', 'output_stride', 'convs_per_block', 'up_interpolate', 'arch', 'stem_patch_kernel', 'stem_patch_stride')} } err={err:.2e}{flag}")
from sleap_nn_amd import _lib as L

print("kernel families over the sweep:", {L.KV_NAMES[c].split(" ")[0]: n for c, n in sorted(kinds_seen.items()) if c})
print("worst", worst)
sys.exit(0 if worst <= 1e-4 else 1)
