"""Wave-private F(2x2,3x3) kernel (Cout 32, Cin 16 / 32) against the oracle and the F(2,3) kernel: python tools/w16_check.py [speed]"""
import sys, time
sys.path.insert(0, ".")
import torch
from oracle import cpu_ref as O
from sleap_nn_amd.architectures.model import Model
DEV = "cuda:0"
if "speed" not in sys.argv:
    for hw, ms in (((64, 64), 8), ((36, 44), 8), ((17, 33), 4), ((96, 80), 4), ((130, 70), 4)):
        bb = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": ms, "stem_stride": None, "middle_block": True,
              "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": ms}
        heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": ms}}
        sd = O.init_state(bb, heads, "single_instance", seed=hw[0], head_scale=1.0)
        g = torch.Generator().manual_seed(hw[1])
        img = torch.randint(0, 256, (3, 1, hw[0], hw[1]), dtype=torch.uint8, generator=g)
        ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
        outs = {}
        for name, v in (("w16", 1), ("w1d", 0)):
            m = Model("unet", bb, heads, "single_instance"); m.load_state_dict(sd); m.set_option("conv_w16", v)
            outs[name] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        print(f"hw {hw} ms {ms}: scale {ref.abs().max().item():.3g} |w16-ref| {(outs['w16']-ref).abs().max().item():.3g} |w1d-ref| {(outs['w1d']-ref).abs().max().item():.3g}", flush=True)
import bench
g = torch.Generator().manual_seed(4321)
B = 32 if "speed" in sys.argv else 2
frames = torch.randint(0, 256, (B, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to(DEV)
res = {}
for name, v in (("w16", 1), ("w1d", 0)):
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(DEV).set_option("conv_w16", v)
    for _ in range(3):
        out = m(frames)
    res[name] = {k: t.clone() for k, t in out.items()}
    m.set_profiling(True)
    for _ in range(8):
        m(frames)
    ms, n = m.read_profile(); m.set_profiling(False)
    labels = [o.label.split(".")[-1].replace("stack0_", "") for o in m.ops]
    print(f"{name} B={B}: forward {sum(ms)/n:.3f} ms | " + " ".join(f"{l}={x/n:.3f}" for l, x in list(zip(labels, ms))[:4]), flush=True)
for k in res["w16"]:
    print(k, "max |w16 - w1d|", (res["w16"][k] - res["w1d"][k]).abs().max().item(), "scale", res["w1d"][k].abs().max().item())
