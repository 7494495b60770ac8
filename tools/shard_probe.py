"""cfg3 forward at the per-rank batches of a strong-scaling run (32 / N frames), with 1 .. 4 copies of the network on as many HIP streams (consecutive batches alternate between the
copies: what a rank of bench.py --gpus N does with two).  python tools/shard_probe.py [B ...]"""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend

dev = torch.device("cuda", 0)
Bs = [int(a) for a in sys.argv[1:]] or [16, 8, 4]
copies = []
for k in range(4):
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05)
    copies.append((HipBackend(m, str(dev), use_graph=True), torch.cuda.Stream(dev)))
for B in Bs:
    x = torch.randint(0, 256, (B, 1, 1024, 1024), dtype=torch.uint8, device=dev)
    xs = [be.static_input((B, 1, 1024, 1024)).copy_(x) for be, _ in copies]
    torch.cuda.synchronize()
    for n in (1, 2, 3, 4):
        def run(steps):
            for i in range(steps):
                be, st = copies[i % n]
                with torch.cuda.stream(st):
                    be(xs[i % n])
        run(8)
        torch.cuda.synchronize()
        t = time.perf_counter()
        steps = 60
        run(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        print(f"B={B} copies={n}: {dt * 1e6:8.1f} us per batch = {B / dt:7.1f} frames/s", flush=True)
