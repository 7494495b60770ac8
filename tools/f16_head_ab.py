"""A/B of the head fused into the fp16 conv epilogue (handle option head_fuse) on the hipGraph forward of (a) the cfg3 network and (b) the cfg5 network (17-channel confidence-map head
behind the 64-channel conv: fused; 4-channel class-map head behind a 128-channel conv: its own launch), 768 x 768 x 16 frames, fp16 pipe; alternating, two rounds, one box."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend

dev = torch.device("cuda", 0)
x = torch.randint(0, 256, (16, 1, 768, 768), dtype=torch.uint8, device=dev)
cfg5_heads = {"confmaps": {"part_names": [f"k{i}" for i in range(17)], "sigma": 2.5, "output_stride": 4, "loss_weight": 1.0},
              "class_maps": {"classes": [f"id{i}" for i in range(4)], "sigma": 12.5, "output_stride": 8, "loss_weight": 1.0}}
for name, heads, mt in (("cfg3 net", bench.CFG3_HEADS, "bottomup"), ("cfg5 net", cfg5_heads, "multi_class_bottomup")):
    for rep in range(2):
        for fuse in (1, 0):
            m = Model("unet", dict(bench.CFG3_BB), heads, mt).init_xavier_(seed=1234, head_scale=0.05)
            m.set_option("head_fuse", fuse)
            be = HipBackend(m, str(dev), use_graph=True, precision="fp16")
            xs = be.static_input((16, 1, 768, 768)).copy_(x)
            for _ in range(10):
                be(xs)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(100):
                be(xs)
            torch.cuda.synchronize()
            print(name, "head_fuse", fuse, "forward us", round((time.perf_counter() - t) / 100 * 1e6, 1), flush=True)
            del be, m
