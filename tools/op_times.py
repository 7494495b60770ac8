"""Per-op HIP-event times of the cfg3 forward under handle options given on the command line (name=value ...)."""
import sys
sys.path.insert(0, ".")
import torch
import bench
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model

B, CFG = None, "cfg3"
opts = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "B":
        B = int(v)
    elif k == "cfg":
        CFG = v
    else:
        opts[k] = int(v)
dev = torch.device("cuda", 0)
if CFG == "cfg4":
    B, S = B or 64, 384
    m = Model("convnext", bench.CFG4_BB, bench.CFG4_HEADS, "centered_instance").init_xavier_(seed=1234, head_scale=0.05).to(dev)
else:
    B, S = B or 32, 1024
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(dev)
for k, v in opts.items():
    m.set_option(k, v)
x = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, device=dev)
for _ in range(3):
    m(x)
torch.cuda.synchronize()
m.set_profiling(True)
N = 10
for _ in range(N):
    m(x)
torch.cuda.synchronize()
ms, n = m.read_profile()
m.set_profiling(False)
codes = m.last_kernels()
tab = m.op_table(B, S, S)
tot = 0.0
for r, t, c in zip(tab, ms, codes):
    t /= n
    tot += t
    if t > 0:
        sh = L.KV_MFMA_SHARE.get(c, 0)
        tf = r["flops"] * sh / t / 1e9 if t else 0
        print(f"{r['label']:44s} {L.KV_NAMES.get(c, '-')[:22]:22s} {str(r.get('out_hw')):12s} {t:7.3f} ms  exec {tf:6.1f} TF/s ({tf/157.3:4.2f})")
print(f"options {opts} B={B}: forward {tot:.3f} ms")
