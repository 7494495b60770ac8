import sys, time
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend
dev = torch.device("cuda", 0)
x = torch.randint(0, 256, (16, 1, 768, 768), dtype=torch.uint8, device=dev)
res = {}
for rep in range(2):
    for fuse in (1, 0):
        m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05)
        m.set_option("head_fuse", fuse)
        be = HipBackend(m, str(dev), use_graph=True, precision="fp16")
        xs = be.static_input((16, 1, 768, 768)).copy_(x)
        for _ in range(10): be(xs)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): be(xs)
        torch.cuda.synchronize()
        print("head_fuse", fuse, "forward us", (time.perf_counter() - t) / 100 * 1e6, flush=True)
        del be, m
