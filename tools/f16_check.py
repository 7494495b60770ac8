"""Parity and speed of the fp16-matrix-pipe forward ("split" / "fp16" precisions) against the goldens, the exact path and the oracle.
    python tools/f16_check.py [quick]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from tests import _golden as G
from sleap_nn_amd.architectures.model import Model
DEV = "cuda:0"
for name in ["unet_tiny_interp.npz", "unet_tiny_trans.npz", "unet_tiny_bu13.npz", "unet_tiny_rgb.npz", "ckpt_bottomup.npz", "ckpt_single_instance.npz"]:
    z = G.load(name); cfg = G.config(z)
    for prec in ("exact", "split", "fp16"):
        m = Model("unet", cfg["backbone"], cfg["heads"], cfg["model_type"]); m.load_state_dict(G.weights(z), strict=True); m.to(DEV).set_precision(prec)
        try:
            out = m(torch.from_numpy(z["image"]).squeeze(1).to(DEV)); torch.cuda.synchronize()
        except Exception as e:
            print(name, prec, "FAILED", e); continue
        errs = {k[4:]: float((out[k[4:]].cpu() - torch.from_numpy(z[k])).abs().max()) for k in z.files if k.startswith("out/")}
        scale = {k[4:]: float(np.abs(z[k]).max()) for k in z.files if k.startswith("out/")}
        print(f"{name:28s} {prec:6s}", {k: f"{v:.2e}/{scale[k]:.2f}" for k, v in errs.items()})
if len(sys.argv) > 1: sys.exit(0)
# cfg3 speed + closeness at full size
B = 32
g = torch.Generator().manual_seed(4321)
frames = torch.randint(0, 256, (B, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to(DEV)
outs = {}
for prec in ("exact", "split", "fp16"):
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(DEV).set_precision(prec)
    for _ in range(3): o = m(frames)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): o = m(frames)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    outs[prec] = {k: v.clone() for k, v in o.items()}
    m.set_profiling(True); m(frames); ms, n = m.read_profile(); m.set_profiling(False)
    print(f"cfg3 B={B} {prec}: {dt*1e3:.2f} ms/forward = {B/dt:.0f} frames/s; per-op ms:", [round(x, 3) for x in ms])
for prec in ("split", "fp16"):
    for k in outs["exact"]:
        e = (outs[prec][k] - outs["exact"][k]).abs().max().item(); s = outs["exact"][k].abs().max().item()
        print(f"cfg3 {prec} vs exact {k}: max abs diff {e:.3e} (scale {s:.3e}, rel {e/s:.2e})")
