"""F(2x2,3x3) kernel against the F(2,3)-along-x and direct kernels: python tools/wino2d_check.py [speed]"""
import sys, time
sys.path.insert(0, ".")
import torch
from oracle import cpu_ref as O
from sleap_nn_amd.architectures.model import Model

DEV = "cuda:0"


def net(filters, max_stride, hw, B=2, seed=0, out_stride=None, heads_kind="single_instance"):
    bb = {"in_channels": 1, "kernel_size": 3, "filters": filters, "filters_rate": 2, "max_stride": max_stride, "stem_stride": None, "middle_block": True,
          "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": out_stride or max_stride}
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": out_stride or max_stride}}
    sd = O.init_state(bb, heads, heads_kind, seed=seed, head_scale=1.0)
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (B, 1, hw[0], hw[1]), dtype=torch.uint8, generator=g)
    outs = {}
    for name, opts in (("w2d", {"conv_wino2d": 1}), ("w1d", {"conv_wino2d": 0}), ("direct", {"conv_wino2d": 0, "conv_wino": 0})):
        m = Model("unet", bb, heads, heads_kind)
        m.load_state_dict(sd)
        for k, v in opts.items():
            m.set_option(k, v)
        outs[name] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].float().cpu()
    ref = O.model_forward(sd, bb, heads, heads_kind, img)["SingleInstanceConfmapsHead"]
    sc = ref.abs().max().item()
    print(f"filters {filters} ms {max_stride} os {out_stride} hw {hw}: scale {sc:.3g}  |w2d-ref| {(outs['w2d']-ref).abs().max().item():.3g}  |w1d-ref| {(outs['w1d']-ref).abs().max().item():.3g}  |direct-ref| {(outs['direct']-ref).abs().max().item():.3g}", flush=True)
    return (outs["w2d"] - ref).abs().max().item() / max(sc, 1e-30)


if "speed" not in sys.argv:
    worst = 0.0
    for f, ms, hw, os_ in ((32, 8, (64, 64), None), (64, 8, (72, 52), None), (32, 8, (36, 44), None), (32, 16, (128, 160), 2), (48, 8, (80, 48), 4), (64, 4, (50, 70), None)):
        worst = max(worst, net(f, ms, hw, out_stride=os_))
    print("worst relative", worst)
import bench
g = torch.Generator().manual_seed(4321)
B = 32 if "speed" in sys.argv else 4
frames = torch.randint(0, 256, (B, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to(DEV)
res = {}
for name, v in (("w2d", 1), ("w1d", 0)):
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(DEV).set_option("conv_wino2d", v)
    for _ in range(3):
        out = m(frames)
    res[name] = {k: t.clone() for k, t in out.items()}
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        m(frames)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    m.set_profiling(True)
    for _ in range(5):
        m(frames)
    ms, n = m.read_profile(); m.set_profiling(False)
    labels = [o.label.split(".")[-1].replace("stack0_", "") for o in m.ops]
    print(f"{name} B={B}: {dt*1e3:.3f} ms/forward; conv sum {sum(x for x, o in zip(ms, m.ops) if o.kind == 2)/n:.3f} ms")
    print(" ".join(f"{l}={x/n:.3f}" for l, x in zip(labels, ms)), flush=True)
for k in res["w2d"]:
    d = (res["w2d"][k] - res["w1d"][k]).abs().max().item()
    print(k, "max |w2d - w1d|", d, "scale", res["w1d"][k].abs().max().item())
