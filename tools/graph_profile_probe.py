"""Can the per-op HIP events of ph_model_forward live INSIDE a captured hipGraph (event-record nodes), so that a profiled step replays at graph speed?
python tools/graph_profile_probe.py   (GPU box, repo root)"""
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend

dev = torch.device("cuda", 0)
B = 32
model = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(dev)
frames = torch.randint(0, 256, (B, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, device=dev)
plain = HipBackend(model, str(dev), use_graph=True)
plain(frames)
torch.cuda.synchronize()


def timeit(fn, n=20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n


print("graph replay, no events: %.3f ms" % timeit(lambda: plain(frames)))
eager = HipBackend(model, str(dev))
model.set_profiling(True)
print("eager with per-op events: %.3f ms" % timeit(lambda: eager(frames)))
ms, n = model.read_profile()
print("  eager profile: n", n, "sum %.3f ms" % (sum(ms) / max(n, 1)))
model.set_profiling(True)
prof = HipBackend(model, str(dev), use_graph=True)  # a second backend on the same handle: its capture sees profiling on
try:
    prof(frames)
    torch.cuda.synchronize()
    print("graph replay with event nodes: %.3f ms" % timeit(lambda: prof(frames)))
    ms, n = model.read_profile()
    print("  graph profile: n", n, "sum %.3f ms" % (sum(ms) / max(n, 1)), "first ops", [round(v / max(n, 1), 4) for v in ms[:6]])
except Exception as e:  # noqa: BLE001
    print("capture with events failed:", repr(e)[:300])
