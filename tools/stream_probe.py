"""How many HIP streams the process made before a two-stream Predictor decides whether its two lanes share a hardware queue (python tools/stream_probe.py <k>: k dummy streams first,
then bench.published_workload_leg).  Before Predictor picked its lanes by a measured overlap (predictor.concurrent_streams): top-down 10 300 / 9 900 / 10 400 / 5 800 / 9 700 frames/s for
k = 0 .. 4, bottom-up 8 500 at k = 0; with it 10 000 - 11 000 and 12 500 - 12 900 for every k."""
import sys, os
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
k = int(sys.argv[1])
keep = []
for i in range(k):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        t = torch.ones(1024, device=dev) * 2
    keep.append((s, t))
torch.cuda.synchronize()
pw = bench.published_workload_leg(200, dev)
print("streams before:", k, "topdown e2e", round(pw["topdown"]["end_to_end_fps"]), "bottomup e2e", round(pw["end_to_end"]["value"]), "single", round(pw["single_instance"]["end_to_end_fps"]), flush=True)
