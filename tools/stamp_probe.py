"""Diagnostic (build with PH_EXTRA_HIPCC_FLAGS=-DPH_STAMP): where do the cycles of one conv3x3
workgroup go?  Runs a 2-op program (input conv 1->C, conv C->C) and prints per-wave cycle shares."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.architectures.unet import OpSpec

def run(cin, cout, hw, batch=32):
    bb = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 2, "stem_stride": None, "middle_block": True, "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    m = Model("unet", bb, {"confmaps": {"part_names": ["a"], "output_stride": 2}}, "single_instance")
    m.ops = [OpSpec(L.OP_INPUT_CONV, -1, -1, 0, 1, 0, cin, 3, L.FLAG_RELU, "w0", "b0", label="in"),
             OpSpec(L.OP_CONV, 0, -1, 1, cin, 0, cout, 3, L.FLAG_RELU, "w1", "b1", label="conv"),
             OpSpec(L.OP_HEAD, 1, -1, -1, cout, 0, 1, 1, 0, "w2", "b2", out_index=0, label="head")]
    m.param_shapes = {"w0": (cin, 1, 3, 3), "b0": (cin,), "w1": (cout, cin, 3, 3), "b1": (cout,), "w2": (1, cout, 1, 1), "b2": (1,)}
    m._state = {k: torch.randn(v) * 0.05 for k, v in m.param_shapes.items()}
    m.backbone.n_slots = 2
    m.heads = m.heads[:1]
    x = torch.randint(0, 256, (batch, 1, hw, hw), dtype=torch.uint8).cuda()
    m.to("cuda:0")(x); torch.cuda.synchronize()
    W = 32768  # words per op (posehip.h: ph_model_set_clock_probe); the conv is op 1 of this program
    buf = torch.zeros(W * len(m.ops), dtype=torch.int64, device="cuda")
    L.check(L.lib().ph_model_set_clock_probe(m._handle, C.c_void_p(buf.data_ptr())))
    m.set_profiling(True)
    for _ in range(5): m(x)
    ms, n = m.read_profile()
    torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(len(m.ops), W)[1].reshape(-1, 4)
    b = b[b.sum(1) > 0]
    import os
    if "STAMP" not in os.environ.get("PH_BUILD", ""):
        nchunks = (cin + 15) // 16
        fl = 2.0 * cin * cout * 9 * hw * hw * batch
        t_ms = ms[1] / n
        clk = np.median(b[:, 1] / np.maximum(b[:, 2], 1)) * 0.1
        span = (b[:, 3].max() + b[np.argmax(b[:, 3]), 1] - b[:, 3].min())
        tot = b[:, 1].astype(np.float64)
        start = b[:, 3].astype(np.float64)
        # keep only the waves of the LAST launch (largest start times)
        order = np.argsort(start)
        nb = 4 * ((hw + 31) // 32) * ((hw + 7) // 8) * batch * ((cout + 63) // 64 if cout >= 64 else (cout + 31) // 32)
        sel = order[-nb:]
        st, tt = start[sel], tot[sel]
        print(f"   last launch: waves {len(sel)}; total cyc p5 {np.percentile(tt,5):.0f} p50 {np.percentile(tt,50):.0f} p95 {np.percentile(tt,95):.0f} max {tt.max():.0f};"
              f" kernel span (first start -> last end) {((st+tt).max()-st.min()):.0f} cyc = {((st+tt).max()-st.min())/clk/1e6:.3f} ms;"
              f" start spread p50 {np.percentile(st-st.min(),50):.0f} max {(st-st.min()).max():.0f}")
        print(f"conv {cin}->{cout} @{hw}^2 x{batch}: {t_ms:.3f} ms = {fl/(t_ms*1e-3)/1e12:.1f} TFLOP/s; waves {len(b)}; clock {clk:.3f} GHz; "
              f"per wave: loop {np.median(b[:,0]):.0f} cyc, total {np.median(b[:,1]):.0f} cyc (epilogue {np.median(b[:,1]-b[:,0]):.0f}); "
              f"ideal mfma/wave {nchunks*18432}; first-start..last-end span {span} cyc = {span/clk/1e6:.3f} ms; sum(total)/4waves/256CU/2 = {b[:,1].sum()/4/512/clk/1e6:.3f} ms")
        return
    tot = b.sum(1)
    nchunks = (cin + 15) // 16
    fl = 2.0 * cin * cout * 9 * hw * hw * batch
    print(f"conv {cin}->{cout} @{hw}^2 x{batch}: {ms[1]/n:.3f} ms = {fl/(ms[1]/n*1e-3)/1e12:.1f} TFLOP/s; waves {len(b)}; cycles/wave median {np.median(tot):.0f}"
          f" | per chunk: commit {np.median(b[:,0])/nchunks:.0f} bar1 {np.median(b[:,1])/nchunks:.0f} mfma {np.median(b[:,2])/nchunks:.0f} bar2 {np.median(b[:,3])/nchunks:.0f} (ideal mfma 18432 x blocks/CU)")

if __name__ == "__main__":
    run(256, 256, 64)
    run(64, 64, 256)
    run(768, 256, 64)
