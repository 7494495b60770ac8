"""cfg1 / cfg2 of BASELINE.json (single-instance UNet f16/r2/max_stride16/output_stride2): layer throughput, eager vs hipGraph.

    python tools/cfg2_bench.py [size] [batch] [n_nodes]
"""
import sys, time
sys.path.insert(0, ".")
import torch
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend
from sleap_nn_amd.inference.layers import SingleInstanceLayer, PostprocessConfig

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 13
bb = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 16, "stem_stride": None, "middle_block": True, "up_interpolate": True,
      "stacks": 1, "convs_per_block": 2, "output_stride": 2}
heads = {"confmaps": {"part_names": [str(i) for i in range(K)], "output_stride": 2}}
m = Model("unet", bb, heads, "single_instance")
m.init_xavier_(seed=1234, head_scale=0.05)
frames = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8).cuda()
gflop = sum(r["flops"] for r in m.op_table(1, S, S)) / 1e9
for graph in (False, True):
    layer = SingleInstanceLayer(HipBackend(m, "cuda:0", use_graph=graph), 2, max_stride=16, postprocess_config=PostprocessConfig(peak_threshold=0.0))
    for _ in range(5):
        out = layer.predict(frames)
    torch.cuda.synchronize()
    n = 50
    t = time.perf_counter()
    for _ in range(n):
        out = layer.predict(frames)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"{S}x{S} B={B} K={K} graph={graph}: {dt*1e3:.3f} ms/batch = {B/dt:.0f} frames/s ({gflop:.2f} GFLOP/frame -> {gflop*B/dt/1e3:.1f} TFLOP/s direct-equivalent)")
