"""Shader clock held during each conv launch of the cfg3 forward (fp16-pipe precisions): python tools/clockprobe_f16.py [split|fp16]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import torch, numpy as np, bench
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd import _lib as L
prec = sys.argv[1] if len(sys.argv) > 1 else "split"
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to("cuda:0").set_precision(prec)
x = torch.randint(0, 256, (32, 1, 1024, 1024), dtype=torch.uint8).cuda()
m(x); torch.cuda.synchronize()
W = 32768  # words per op (posehip.h: ph_model_set_clock_probe)
buf = torch.zeros(W * len(m.ops), dtype=torch.int64, device="cuda")
L.check(L.lib().ph_model_set_clock_probe(m._handle, C.c_void_p(buf.data_ptr())))
for _ in range(10): m(x)
torch.cuda.synchronize()
b = buf.cpu().numpy().reshape(len(m.ops), W)[:, : 2 * 1024].reshape(len(m.ops), 1024, 2)
for i, op in enumerate(m.ops):
    nz = b[i, :, 1] > 0
    if nz.sum() == 0: continue
    clk = b[i, nz, 0] / b[i, nz, 1] * 0.1
    print(f"{op.label.split('.')[-1]:45s} wgs {nz.sum():4d} clock GHz median {np.median(clk):.3f} (p10 {np.percentile(clk,10):.3f} p90 {np.percentile(clk,90):.3f}); wg lifetime median {np.median(b[i,nz,1])/100:.1f} us")
