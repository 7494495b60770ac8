"""F(4x4,3x3) kernel (conv3x3_wino4_kernel, with and without the folded bilinear x2) against the oracle and the F(2x2,3x3) kernel:
    python tools/w4_check.py [speed]"""
import sys, time
sys.path.insert(0, ".")
import torch
from oracle import cpu_ref as O
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model

DEV = "cuda:0"


def net(filters, max_stride, hw, B=2, seed=0, out_stride=None, min_cin=64):
    os_ = out_stride or max_stride
    bb = {"in_channels": 1, "kernel_size": 3, "filters": filters, "filters_rate": 2, "max_stride": max_stride, "stem_stride": None, "middle_block": True,
          "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": os_}
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": os_}}
    sd = O.init_state(bb, heads, "single_instance", seed=seed, head_scale=1.0)
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (B, 1, hw[0], hw[1]), dtype=torch.uint8, generator=g)
    collect = {}
    ref = O.model_forward(sd, bb, heads, "single_instance", img, collect=collect)["SingleInstanceConfmapsHead"]
    outs, kinds = {}, {}
    for name, opts, keep in (("w4fold", {"conv_wino4": 3, "conv_wino4_min_cin": min_cin}, False), ("w4", {"conv_wino4": 3, "conv_wino4_min_cin": min_cin}, True), ("w2d", {"conv_wino4": 0}, False)):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        for k, v in opts.items():
            m.set_option(k, v)
        m.to(DEV).set_keep_activations(keep)
        outs[name] = m(img.to(DEV))["SingleInstanceConfmapsHead"].float().cpu()
        kv = m.last_kernels()
        kinds[name] = sum(1 for c in kv if c == L.KV_WINO4)
        if keep:
            worst = ("", 0.0)
            for lab, t in collect.items():
                if lab not in m.backbone.labels:
                    continue
                try:
                    got = m.read_activation(lab, t.shape[0], t.shape[-2:]).cpu()
                except KeyError:
                    continue
                e = (got - t).abs().max().item() / max(t.abs().max().item(), 1e-30)
                if e > worst[1]:
                    worst = (lab, e)
            print("   worst activation (w4, no fold):", worst)
    sc = ref.abs().max().item()
    print(f"filters {filters} ms {max_stride} os {os_} hw {hw}: scale {sc:.3g} wino4 launches {kinds}  |w4fold-ref| {(outs['w4fold']-ref).abs().max().item()/sc:.3g}  |w4-ref| {(outs['w4']-ref).abs().max().item()/sc:.3g}  "
          f"|w2d-ref| {(outs['w2d']-ref).abs().max().item()/sc:.3g} (relative)", flush=True)
    return max((outs[k] - ref).abs().max().item() for k in ("w4fold", "w4")) / max(sc, 1e-30)


if "speed" not in sys.argv:
    worst = 0.0
    for f, ms, hw, os_ in ((32, 8, (128, 128), None), (16, 32, (128, 192), 4), (32, 16, (128, 160), 2), (64, 8, (72, 52), None), (64, 4, (48, 80), None), (32, 8, (36, 44), None)):
        worst = max(worst, net(f, ms, hw, out_stride=os_))
    print("worst relative", worst)
import bench
g = torch.Generator().manual_seed(4321)
B = 32 if "speed" in sys.argv else 4
frames = torch.randint(0, 256, (B, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to(DEV)
res = {}
for name, opts in (("w4", {"conv_wino4": 1}), ("w4_all128", {"conv_wino4": 1, "conv_wino4_min_cin": 128}), ("w2d", {"conv_wino4": 0})):
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(DEV)
    for k, v in opts.items():
        m.set_option(k, v)
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
    kv = m.last_kernels()
    labels = [o.label.split(".")[-1].replace("stack0_", "") for o in m.ops]
    print(f"{name} B={B}: {dt*1e3:.3f} ms/forward; conv sum {sum(x for x, o in zip(ms, m.ops) if o.kind == 2)/n:.3f} ms")
    print(" ".join(f"{l}[{c}]={x/n:.3f}" for l, x, c in zip(labels, ms, kv)), flush=True)
for k in res["w2d"]:
    for name in ("w4", "w4_all128"):
        d = (res[name][k] - res["w2d"][k]).abs().max().item()
        print(k, name, "max |w4 - w2d|", d, "scale", res["w2d"][k].abs().max().item(), "relative", d / res["w2d"][k].abs().max().item())
