"""conv3x3_f16_rows_kernel against conv3x3_f16_persist_kernel on whole networks (plain fp16): head outputs with "conv_f16_rows" 0 / 2, and per-op times of both.
   python tools/f16_rows_check.py [B] [S] [time]"""
import sys
sys.path.insert(0, ".")
import torch
import bench
from sleap_nn_amd.architectures.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 768
timing = len(sys.argv) > 3
modes = (2,) if (len(sys.argv) > 3 and sys.argv[3] in ('only2',)) else (1,) if (len(sys.argv) > 3 and sys.argv[3] == 'pmc') else ((1,) if (len(sys.argv) > 3 and sys.argv[3] == 'only1') else (0, 2, 1))
if len(sys.argv) > 3 and sys.argv[3] == 'pmc':
    timing = False
dev = torch.device("cuda", 0)
heads = {"confmaps": {"part_names": [f"k{i}" for i in range(17)], "sigma": 2.5, "output_stride": 4, "loss_weight": 1.0},
         "class_maps": {"classes": [f"id{i}" for i in range(4)], "sigma": 12.5, "output_stride": 8, "loss_weight": 1.0}}
m = Model("unet", dict(bench.CFG3_BB), heads, "multi_class_bottomup").init_xavier_(seed=1234, head_scale=1.0).to(dev).set_precision("fp16")
x = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, device=dev)
for kv in sys.argv[4:]:
    if '=' in kv:
        m.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    else:
        m.set_option('upsample_f16math', float(kv))
outs = {}
for mode in modes:
    m.set_option("conv_f16_rows", mode)
    o = m(x)
    torch.cuda.synchronize()
    outs[mode] = {k: v.clone() for k, v in o.items()}
    print("mode", mode, "kernels", m.last_kernels() if hasattr(m, "last_kernels") else "", flush=True)
for k in (outs[0] if 0 in outs else {}):
    sc = outs[0][k].abs().max().item()
    for mode in (2, 1):
        d = (outs[mode][k] - outs[0][k]).abs().max().item()
        print(f"{k}: scale {sc:.4g}  max |rows({mode}) - persist| {d:.3g}  finite {bool(torch.isfinite(outs[mode][k]).all())}", flush=True)
if timing:
    for mode in modes:
        m.set_option("conv_f16_rows", mode)
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
        print(f"--- conv_f16_rows = {mode}")
        for r, t in zip(tab, ms):
            t /= n
            tot += t
            if t > 0:
                print(f"{r['label']:44s} {str(r.get('out_hw')):12s} {t*1e3:8.1f} us  {r['flops']/t/1e9:8.1f} TF/s direct", flush=True)
        print(f"fp16 B={B} {S}x{S} mode {mode}: forward {tot*1e3:.1f} us (per-op events)", flush=True)
        g = torch.cuda.CUDAGraph()
        m.set_keep_activations(False) if hasattr(m, "set_keep_activations") else None
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        t0.record()
        for _ in range(50):
            m(x)
        t1.record()
        torch.cuda.synchronize()
        print(f"mode {mode}: eager back-to-back forward {t0.elapsed_time(t1)/50*1e3:.1f} us", flush=True)
