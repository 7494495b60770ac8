"""Per-op times of the cfg3 forward, the two 1x1 heads in particular: python tools/head_probe.py"""
import sys
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
DEV = "cuda:0"
g = torch.Generator().manual_seed(4321)
frames = torch.randint(0, 256, (32, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to(DEV)
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(DEV)
for a in sys.argv[1:]:
    k, v = a.split("=")
    m.set_option(k, float(v))
for _ in range(3):
    m(frames)
m.set_profiling(True)
for _ in range(10):
    m(frames)
ms, n = m.read_profile(); m.set_profiling(False)
tab = m.op_table(32, bench.SIZE, bench.SIZE)
for i, (t, r) in enumerate(zip(ms, tab)):
    if i >= len(ms) - 4:
        print(f"op {i:2d} {r.get('label', r.get('name', '?'))!s:40s} {t / n:.4f} ms  cin {r.get('cin0')} cout {r.get('cout')}")
print(f"forward {sum(ms) / n:.3f} ms")
