"""Max |GPU - oracle| on the cfg3 network's heads at a reduced frame size (the oracle runs on the CPU):
    python tools/parity_margin.py [size] [batch]       (PH_CONV_WINO=0 for the direct 9-tap kernel)"""
import sys

sys.path.insert(0, ".")
import torch

import bench
from oracle import cpu_ref as O
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup")
m.init_xavier_(seed=1234, head_scale=0.05)
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(4321)
img = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, generator=g)
ref = O.model_forward(sd, bench.CFG3_BB, bench.CFG3_HEADS, "bottomup", img)
out = HipBackend(m, "cuda:0")(img)
for k, v in ref.items():
    d = (out[k].cpu() - v).abs()
    print(f"{k:32s} max|ref| {float(v.abs().max()):.4f}  max err {float(d.max()):.3e}  mean err {float(d.mean()):.3e}")
