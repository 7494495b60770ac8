"""Shader clock held by every F(4x4,3x3) launch of the cfg3 forward (exact fp32), per layer: needs a library built with -DW4_CLOCK
(PH_EXTRA_HIPCC_FLAGS=-DW4_CLOCK python -m sleap_nn_amd.build --force; rebuild without it afterwards).  python tools/clockprobe_layers.py [B [forwards]]"""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch, bench
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = 32768  # words per op (posehip.h: ph_model_set_clock_probe)
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to("cuda:0")
x = torch.randint(0, 256, (B, 1, 1024, 1024), dtype=torch.uint8).cuda()
m(x)
torch.cuda.synchronize()
buf = torch.zeros(W * len(m.ops), dtype=torch.int64, device="cuda")
L.check(L.lib().ph_model_set_clock_probe(m._handle, C.c_void_p(buf.data_ptr())))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):  # (the record holds the LAST forward: enough forwards for the power management to settle)
    m(x)
torch.cuda.synchronize()
L.check(L.lib().ph_model_set_clock_probe(m._handle, None))
codes = m.last_kernels()
b = buf.cpu().numpy().reshape(len(m.ops), W)
tab = m.op_table(B, 1024, 1024)
for i, (row, code) in enumerate(zip(tab, codes)):
    if code != L.KV_WINO4:
        continue
    r = b[i, : 2 * 256].reshape(256, 2)
    nz = r[:, 1] > 0
    if nz.sum() == 0:
        print(f"{row['label']:44s} no record (library not built with -DW4_CLOCK?)")
        continue
    clk = r[nz, 0] / r[nz, 1] * 0.1
    print(f"{row['label']:44s} {str(row.get('out_hw')):12s} workgroups {int(nz.sum()):4d}  clock GHz median {np.median(clk):.3f}  p10 {np.percentile(clk, 10):.3f}  p90 {np.percentile(clk, 90):.3f}   kernel {np.median(r[nz, 1]) / 100:.0f} us")
