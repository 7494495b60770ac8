"""The 8- / 4-frame per-rank shards of cfg3 as whole pipelined steps with 2 or 3 copies of the network at 4 frames (bench.SHARD_LANES_4): python tools/shard_lanes_ab.py"""
import sys, json, subprocess
for n in (2, 3):
    code = f"import bench,sys; bench.SHARD_LANES_4={n}; sys.argv=['bench.py','--no-cpu-baseline','--no-alt-precisions','--no-h2d-leg','--legs-file','gpurun_out/tmp_legs.json']; bench.main()"
    subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    d = json.load(open("gpurun_out/tmp_legs.json"))
    print("lanes at 4 frames:", n, {k: (round(v["value"]), round(v["two_streams"]["value"]), v["two_streams"].get("copies")) for k, v in d["strong_scaling_shards"].items() if isinstance(v, dict) and "two_streams" in v}, flush=True)
