"""3x3 weight gradients in the Winograd F(2x2,3x3) domain (wgrad_wino = 1) against the direct nine-tap kernel (0): every parameter
gradient of one backward at several shapes (odd sizes, narrow maps, concat layers), then cfg3 step timing.  python tools/wgrad_check.py"""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from oracle import cpu_ref as O
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.training.module import TrainingModule

DEV = "cuda:0"
NODES = ["a", "b", "c"]
for filters, hw, B in ((16, (64, 96), 2), (16, (80, 112), 3), (32, (48, 80), 2), (24, (112, 48), 1)):
    bb = {"in_channels": 1, "kernel_size": 3, "filters": filters, "filters_rate": 2, "max_stride": 16, "stem_stride": None, "middle_block": True,
          "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    heads = {"confmaps": {"part_names": NODES, "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0}}
    g = torch.Generator().manual_seed(hw[0])
    img = torch.randint(0, 256, (B, 1, hw[0], hw[1]), dtype=torch.uint8, generator=g).to(DEV)
    tg = {"SingleInstanceConfmapsHead": torch.rand((B, 3, (hw[0] + 1) // 2, (hw[1] + 1) // 2), generator=g).to(DEV)}
    grads = {}
    for v in (0, 1):
        m = Model("unet", bb, heads, "single_instance").init_xavier_(seed=7, head_scale=1.0)
        tm = TrainingModule(m, DEV, lr=1e-4)
        m.set_option("wgrad_wino", v)
        tm.forward_backward(img, tg)
        grads[v] = tm.grads.clone().cpu()
    d = (grads[1] - grads[0]).abs().max().item()
    print(f"filters {filters} hw {hw} B {B}: |g| max {grads[0].abs().max().item():.3g}  |wino - direct| max {d:.3g}  rel {d / grads[0].abs().max().item():.2g}", flush=True)

B = 32
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (B, 1, 1024, 1024), dtype=torch.uint8, generator=g).to(DEV)
tg = {"MultiInstanceConfmapsHead": torch.rand((B, 13, 256, 256), generator=g).to(DEV), "PartAffinityFieldsHead": torch.rand((B, 24, 128, 128), generator=g).to(DEV)}
res = {}
for v in (1, 0, 1, 0):
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05)
    tm = TrainingModule(m, DEV, lr=1e-4)
    m.set_option("wgrad_wino", v)
    for _ in range(2):
        tm.training_step({"image": img, **tg})
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        tm.training_step({"image": img, **tg})
    torch.cuda.synchronize()
    print(f"cfg3 B=32 wgrad_wino {v}: step {(time.perf_counter() - t) / 5 * 1e3:.2f} ms", flush=True)
    tm.forward_backward(img, tg)
    res[v] = tm.grads.clone()
    del tm, m
d = (res[1] - res[0]).abs().max().item()
print(f"cfg3 grads: |g| max {res[0].abs().max().item():.3g} |wino - direct| {d:.3g}")
