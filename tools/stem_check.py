"""Fused stem variants (stem_wino 2 / 1 / 0) against each other and the oracle: python tools/stem_check.py"""
import sys
sys.path.insert(0, ".")
import torch
from oracle import cpu_ref as O
from sleap_nn_amd.architectures.model import Model
DEV = "cuda:0"
for cin, hw in ((1, (64, 64)), (1, (37, 45)), (3, (40, 72)), (1, (130, 70))):
    bb = {"in_channels": cin, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 2, "stem_stride": None, "middle_block": True,
          "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": 2}}
    sd = O.init_state(bb, heads, "single_instance", seed=hw[0], head_scale=1.0)
    g = torch.Generator().manual_seed(hw[1])
    img = torch.randint(0, 256, (3, cin, hw[0], hw[1]), dtype=torch.uint8, generator=g)
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    outs = {}
    for v in (2, 1, 0):
        m = Model("unet", bb, heads, "single_instance"); m.load_state_dict(sd); m.set_option("stem_wino", v)
        outs[v] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
    print(f"cin {cin} hw {hw}: scale {ref.abs().max().item():.3g} " + " ".join(f"|stem{v}-ref| {(outs[v]-ref).abs().max().item():.3g}" for v in (2, 1, 0)), flush=True)
import bench
g = torch.Generator().manual_seed(4321)
frames = torch.randint(0, 256, (32, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to(DEV)
for v in (2, 1):
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(DEV).set_option("stem_wino", v)
    for _ in range(3):
        m(frames)
    m.set_profiling(True)
    for _ in range(8):
        m(frames)
    ms, n = m.read_profile(); m.set_profiling(False)
    print(f"stem_wino {v}: forward {sum(ms)/n:.3f} ms, stem {ms[0]/n:.3f} ms", flush=True)
