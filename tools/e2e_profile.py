"""cProfile of Predictor.predict on the published bottom-up workload (100 host frames, batch 4): where the host time of the pipelined path goes (development probe)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sleap_nn_amd.inference.predictor import Predictor

root = os.path.join(ROOT, "tests", "golden", "ckpt_dirs", "minimal_instance_bottomup")
z = np.load(os.path.join(ROOT, "tests", "golden", "ckpt_bottomup.npz"), allow_pickle=False)
two = torch.from_numpy(z["image"]).squeeze(1)
vid = torch.cat([two, two.flip(-1)], 0)[:, :, 32:352, :].repeat(25, 1, 1, 2)[..., :560].contiguous()
pred = Predictor.from_model_paths([root], device="cuda:0", batch_size=4, peak_threshold=0.2)
pred.predict(vid[:16])
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    outs = pred.predict(vid)
    dt = time.perf_counter() - t0
    print(f"pass {rep}: {dt * 1e3:.1f} ms for {vid.shape[0]} frames = {vid.shape[0] / dt:.0f} frames/s", flush=True)
pr = cProfile.Profile()
pr.enable()
pred.predict(vid)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 35)

# the same with the worker's share: wall time of the main loop's pieces
import threading

layer = pred.layer
orig_finish = layer._finish_packed
acc = {"finish": 0.0, "n": 0, "sync": 0.0}


def timed_finish(h):
    t0 = time.perf_counter()
    h["event"].synchronize()
    t1 = time.perf_counter()
    o = orig_finish(h)
    acc["sync"] += t1 - t0
    acc["finish"] += time.perf_counter() - t1
    acc["n"] += 1
    return o


layer._finish_packed = timed_finish
t0 = time.perf_counter()
pred.predict(vid)
dt = time.perf_counter() - t0
print(f"pass with timed worker: {dt * 1e3:.1f} ms; worker: event wait {acc['sync'] * 1e3:.1f} ms, finish {acc['finish'] * 1e3:.1f} ms over {acc['n']} batches")
