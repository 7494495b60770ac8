import sys, time
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend
dev = torch.device("cuda", 0)
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05)
be = HipBackend(m, str(dev), use_graph=True)
for B in (32, 4):
    x = torch.randint(0, 256, (B, 1, 1024, 1024), dtype=torch.uint8, device=dev)
    xs = be.static_input((B, 1, 1024, 1024)).copy_(x)
    for _ in range(5): be(xs)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): be(xs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print(B, "graph forward us", dt * 1e6, "kernels", [L.KV_NAMES[c].split(" ")[0][8:] for c in m.last_kernels() if c not in (0, 11)])
