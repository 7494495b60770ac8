import sys, ctypes as C
sys.path.insert(0, '.')
import torch, numpy as np
import bench
from oracle import cpu_ref as O
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd import _lib as L
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup"); m.load_state_dict(O.init_state(bench.CFG3_BB, bench.CFG3_HEADS, "bottomup"))
x = torch.randint(0, 256, (32, 1, 1024, 1024), dtype=torch.uint8).cuda()
m.to("cuda:0")(x); torch.cuda.synchronize()
W = 32768  # words per op (posehip.h: ph_model_set_clock_probe)
buf = torch.zeros(W * len(m.ops), dtype=torch.int64, device="cuda")
L.check(L.lib().ph_model_set_clock_probe(m._handle, C.c_void_p(buf.data_ptr())))
for _ in range(20): m(x)
torch.cuda.synchronize()
# every op has its own record block; kernels that write {d memtime, d memrealtime} pairs per workgroup (tools/clockprobe_layers.py reads them per layer)
b = buf.cpu().numpy().reshape(len(m.ops), W)[:, : 2 * 1024].reshape(-1, 2)
nz = b[:, 1] > 0
clk = b[nz, 0] / b[nz, 1] * 100e6 / 1e9
print("blocks", nz.sum(), "clock GHz median", np.median(clk), "p10", np.percentile(clk, 10), "p90", np.percentile(clk, 90))
print("memtime cycles per block: median", np.median(b[nz,0]), "max", b[nz,0].max())
