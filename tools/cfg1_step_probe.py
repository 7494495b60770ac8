"""BASELINE cfg1 as a synchronous step (InferenceLayer.predict_graphed + a host sync per frame): what the GPU runs per step under
`rocprofv3 --kernel-trace -- python3 tools/cfg1_step_probe.py` + tools/trace_gaps.py -- the forward's 15 launches, global_peaks_kernel, one elementwise
kernel (undo_stride) and ~35 - 48 us of launch + sync latency between steps."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend
from sleap_nn_amd.inference.layers import PostprocessConfig, SingleInstanceLayer
dev = torch.device("cuda", 0)
heads = {"confmaps": {"part_names": [f"k{i}" for i in range(5)], "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0}}
model = Model("unet", bench.SI_BB, heads, "single_instance").init_xavier_(seed=1234, head_scale=0.05).to(dev)
backend = HipBackend(model, str(dev), use_graph=True)
layer = SingleInstanceLayer(backend, 2, max_stride=16, postprocess_config=PostprocessConfig(peak_threshold=0.0))
frames = torch.randint(0, 256, (1, 1, 256, 256), dtype=torch.uint8, device=dev)
g = layer.graph_input((1, 1, 256, 256)).copy_(frames)
for _ in range(20):
    o = layer.predict_graphed(g)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    o = layer.predict_graphed(g)
    torch.cuda.synchronize()
print("sync step us", (time.perf_counter() - t) / 200 * 1e6)
# host cost of one call (enqueue only): the loop returns long before the GPU is done
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    o = layer.predict_graphed(g)
host = (time.perf_counter() - t) / 200 * 1e6
torch.cuda.synchronize()
print("host us per call (enqueue only)", host)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    o = layer.predict_graphed(g)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
# the same step launched kernel by kernel (InferenceLayer.predict: preprocessing, one ctypes call for the forward's launches, the post-process ops), synchronous
for fn, name in ((lambda: layer.predict(frames), "predict (eager launches)"), (lambda: layer.predict_graphed(g), "predict_graphed (own buffer)")):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(300):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    ts.sort()
    print(name, "sync step us: median", ts[150] * 1e6, "p10", ts[30] * 1e6)
