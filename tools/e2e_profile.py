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

# GPU time of the pieces of one pipelined batch (HIP events, 200 repetitions each)
layer._finish_packed = orig_finish
dev = torch.device("cuda:0")
batch = vid[:4].pin_memory()
bd = batch.to(dev)
h = layer._enqueue_scoring_graphed(bd)
torch.cuda.synchronize()
entry = [e for e in layer._step_graphs.values() if e[1].shape == bd.shape][0]
graph, static_in, packed = entry[0], entry[1], entry[2]
host = torch.empty(packed.numel(), dtype=torch.float32, pin_memory=True)


def gpu_us(fn, n=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print(f"graph replay (resize + pad + forward at {tuple(entry[4].processed_size)} + peaks + PAF scoring): {gpu_us(graph.replay):.1f} us")
print(f"H2D of the batch ({batch.numel() / 1e3:.0f} KB, pinned -> graph input): {gpu_us(lambda: static_in.copy_(batch, non_blocking=True)):.1f} us")
print(f"D2H of the arena ({packed.numel() * 4 / 1e3:.0f} KB): {gpu_us(lambda: host.copy_(packed, non_blocking=True)):.1f} us")
xp, info = layer.preprocess(bd)
be = layer.backend
print(f"preprocess alone (resize + pad): {gpu_us(lambda: layer.preprocess(bd)):.1f} us (host-bound if > the kernels)")
fg = torch.cuda.CUDAGraph()
with torch.cuda.graph(fg):
    raw = be.model.forward(xp.squeeze(1))
print(f"forward alone at {tuple(xp.shape)} (graph): {gpu_us(fg.replay):.1f} us")
