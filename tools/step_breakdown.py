"""Host-side time of the bench step's three parts (enqueue, finish of the previous batch, wait for the grouping worker)."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from concurrent.futures import ThreadPoolExecutor
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend
from sleap_nn_amd.inference.layers import BottomUpLayer
from sleap_nn_amd.inference.ops.paf import PAFScorer
from sleap_nn_amd.inference.preprocess_info import PreprocInfo
from sleap_nn_amd.inference.streaming import group_scored_batch

B, SIZE = 32, 1024
dev = torch.device("cuda", 0)
model = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup")
model.init_xavier_(seed=1234, head_scale=0.05)
layer = BottomUpLayer(HipBackend(model, str(dev)), PAFScorer.from_config(bench.CFG3_HEADS), 4, 8, max_stride=32)
frames = torch.randint(0, 256, (B, 1, 1, SIZE, SIZE), dtype=torch.uint8).to(dev)
cms, pafs = bench.rendered_heads(B, dev)
info = PreprocInfo(original_size=(SIZE, SIZE), processed_size=(SIZE, SIZE), eff_scale=torch.ones(B), output_stride=4)
pool = ThreadPoolExecutor(max_workers=1)
params = layer.grouping_params()
inflight, pending = [], []
T = [0.0, 0.0, 0.0, 0.0]
gt = []

def timed_group(sb, params):
    t = time.perf_counter(); r = group_scored_batch(sb, params); gt.append(time.perf_counter() - t); return r

def step(rec):
    t0 = time.perf_counter()
    raw = layer.backend(frames)
    t1 = time.perf_counter()
    inflight.append(layer._enqueue_scoring({"MultiInstanceConfmapsHead": cms, "PartAffinityFieldsHead": pafs}, info))
    t2 = time.perf_counter()
    if len(inflight) > 1:
        pending.append(pool.submit(timed_group, layer._finish_scoring(inflight.pop(0)), params))
    t3 = time.perf_counter()
    out = pending.pop(0).result() if len(pending) > 1 else None
    t4 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): T[i] += d

for _ in range(4): step(False)
torch.cuda.synchronize()
N = 20
t = time.perf_counter()
for _ in range(N): step(True)
torch.cuda.synchronize()
el = time.perf_counter() - t
print(f"step {el/N*1e3:.2f} ms; host: forward enqueue {T[0]/N*1e3:.2f}, scoring enqueue {T[1]/N*1e3:.2f}, finish prev {T[2]/N*1e3:.2f}, wait worker {T[3]/N*1e3:.2f}; grouping worker {sum(gt)/len(gt)*1e3:.2f} ms/batch")
