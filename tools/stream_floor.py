"""What a 109-MB read costs on this GPU with library kernels (the floor the one-pass peak kernel is compared against): torch reductions / copies of the cfg3 confidence maps."""
import sys

sys.path.insert(0, ".")
import torch

import bench

dev = torch.device("cuda", 0)
cms, _ = bench.rendered_heads(32, dev)
nbytes = cms.numel() * 4
big = torch.empty(8 * cms.numel(), device=dev)


def t(label, fn, b):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 50
    print(f"{label:40s} {us:8.1f} us  {b / us / 1e3:8.0f} GB/s ({b / us / 1e3 / 8000:.2f} of 8 TB/s)")


t("amax over the maps (109 MB read)", lambda: cms.amax(), nbytes)
t("sum over the maps (109 MB read)", lambda: cms.sum(), nbytes)
t("(cms > 0.2).any()", lambda: (cms > 0.2).any(), nbytes)
out = torch.empty_like(cms)
t("copy (109 MB read + 109 MB write)", lambda: out.copy_(cms), 2 * nbytes)
t("sum over 872 MB", lambda: big.sum(), 8 * nbytes)
