"""bench.published_workload_leg with n = 1 .. 4 copies of each small network on as many lanes (Predictor.from_model_paths(streams = n) / Predictor(replicas = ...)): python tools/streams_n_probe.py"""
import sys
sys.path.insert(0, ".")
import torch, bench
import benchlegs.published

dev = torch.device("cuda", 0)
for n in (1, 2, 3, 4):
    benchlegs.published.PUBLISHED_LANES = n
    pw = bench.published_workload_leg(200, dev)
    print("lanes", n, "bottom-up e2e", round(pw["end_to_end"]["value"]), "single-instance", round(pw["single_instance"]["end_to_end_fps"]), "top-down", round(pw["topdown"]["end_to_end_fps"]), flush=True)
