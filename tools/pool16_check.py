"""Fused conv + pool with 16 padded output channels on every exact-path kernel family (conv_wino 1 / 0, conv_w16 1 / 0) vs the oracle."""
import sys
sys.path.insert(0, ".")
import torch
from oracle import cpu_ref as O
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model
DEV = "cuda:0"
for filters, hw in ((8, (64, 96)), (8, (40, 72)), (12, (48, 48))):
    bb = {"in_channels": 1, "kernel_size": 3, "filters": filters, "filters_rate": 2, "max_stride": 8, "stem_stride": None, "middle_block": True,
          "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": 2}}
    sd = O.init_state(bb, heads, "single_instance", seed=filters, head_scale=1.0)
    g = torch.Generator().manual_seed(filters)
    img = torch.randint(0, 256, (2, 1, hw[0], hw[1]), dtype=torch.uint8, generator=g)
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    for wino, w16 in ((1, 1), (1, 0), (0, 1), (0, 0)):
        m = Model("unet", bb, heads, "single_instance"); m.load_state_dict(sd)
        m.set_option("conv_wino", wino); m.set_option("conv_w16", w16)
        fused = [(o.cin0, o.cout) for o in m.ops if o.kind == L.OP_CONV and o.dst2 >= 0]
        out = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        print(f"filters {filters} hw {hw} conv_wino {wino} conv_w16 {w16}: fused conv+pool layers {fused}: max |hip - oracle| {(out - ref).abs().max().item():.3g} (scale {ref.abs().max().item():.3g})", flush=True)
