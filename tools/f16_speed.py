"""Per-op times of the cfg3 forward at a given precision: python tools/f16_speed.py [split|fp16|exact] [batch]"""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
prec = sys.argv[1] if len(sys.argv) > 1 else "split"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator().manual_seed(4321)
frames = torch.randint(0, 256, (B, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to("cuda:0")
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to("cuda:0").set_precision(prec)
for _ in range(3): m(frames)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): m(frames)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
m.set_profiling(True)
for _ in range(5): m(frames)
ms, n = m.read_profile(); m.set_profiling(False)
labels = [o.label.split(".")[-1].replace("stack0_", "") for o in m.ops]
print(f"{prec} B={B}: {dt*1e3:.3f} ms/forward; conv sum {sum(x for x, o in zip(ms, m.ops) if o.kind == 2)/n:.3f} ms")
print(" ".join(f"{l}={x/n:.3f}" for l, x in zip(labels, ms)))
