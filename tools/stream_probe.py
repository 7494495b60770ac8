import sys
sys.path.insert(0, ".")
import torch, bench
import benchlegs.published
dev = torch.device("cuda", 0)
k = int(sys.argv[1]); benchlegs.published.PUBLISHED_LANES = int(sys.argv[2])
keep = []
for i in range(k):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        t = torch.ones(1024, device=dev) * 2
    keep.append((s, t))
torch.cuda.synchronize()
pw = bench.published_workload_leg(200, dev)
print("streams before:", k, "lanes", benchlegs.published.PUBLISHED_LANES, "topdown", round(pw["topdown"]["end_to_end_fps"]), "bottomup", round(pw["end_to_end"]["value"]), "single", round(pw["single_instance"]["end_to_end_fps"]), flush=True)
