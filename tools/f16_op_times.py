"""Per-op HIP-event times of the cfg5-style forward (cfg3 UNet, 768 x 768, batch 16) in the fp16 / split / exact precisions: python tools/f16_op_times.py [fp16|split|exact] [B] [S]"""
import sys
sys.path.insert(0, ".")
import torch
import bench
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 768
dev = torch.device("cuda", 0)
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(dev).set_precision(prec)
x = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, device=dev)
for _ in range(3):
    m(x)
torch.cuda.synchronize()
m.set_profiling(True)
for _ in range(10):
    m(x)
torch.cuda.synchronize()
ms, n = m.read_profile()
m.set_profiling(False)
tab = m.op_table(B, S, S)
tot = 0.0
for r, t in zip(tab, ms):
    t /= n
    tot += t
    if t > 0:
        print(f"{r['label']:44s} {str(r.get('out_hw')):12s} {t*1e3:8.1f} us  {r['flops']/t/1e9:8.1f} TF/s direct   {r['bytes']/t/1e6:8.1f} GB/s algorithmic")
print(f"{prec} B={B} {S}x{S}: forward {tot*1e3:.1f} us (per-op events)")
