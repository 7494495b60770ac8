"""Board power / shader clock while the cfg3 forward loops: python tools/power_probe.py [conv_wino2d 0|1] [seconds]
(rocm-smi polled from a thread; the in-kernel clock of the conv kernels: tools/w2_stamp.py, tools/stamp_probe.py)"""
import subprocess, sys, threading, time
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
g = torch.Generator().manual_seed(4321)
frames = torch.randint(0, 256, (32, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to("cuda:0")
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to("cuda:0").set_option("conv_wino2d", v)
stop = False
samples = []
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
            samples.append([l.strip() for l in o.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "junction", "fclk"))])
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.5)
t = threading.Thread(target=poll); t.start()
n = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < secs:
    for _ in range(10):
        m(frames)
    torch.cuda.synchronize(); n += 10
dt = time.perf_counter() - t0
stop = True; t.join()
print(f"conv_wino2d={v}: {dt / n * 1e3:.3f} ms/forward over {n} forwards")
for s in samples[2::3]:
    print("  ", " | ".join(s))
