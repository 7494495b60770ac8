"""Diagnostic (build with PH_EXTRA_HIPCC_FLAGS=-DPH_W2_STAMP): per-wave cycle shares of conv3x3_wino2d_kernel phases.
python tools/w2_stamp.py  -> K loop / exchange phases / stores per tile, per wave role"""
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
    m.to("cuda:0")
    m.set_option("head_fuse", 0)  # (the conv's own epilogue, not the fused head's 64 extra MFMAs per finishing wave)
    m.set_option("conv_wino4", 0)  # (this kernel, also where the F(4x4,3x3) kernel would take the layer)
    m(x); torch.cuda.synchronize()
    W = 32768  # words per op (posehip.h: ph_model_set_clock_probe); the conv is op 1 of this program
    buf = torch.zeros(W * len(m.ops), dtype=torch.int64, device="cuda")
    L.check(L.lib().ph_model_set_clock_probe(m._handle, C.c_void_p(buf.data_ptr())))
    m.set_profiling(True)
    for _ in range(5): m(x)
    ms, n = m.read_profile()
    torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(len(m.ops), W)[1, : 256 * 8 * 8].reshape(256, 8, 8).astype(np.float64)
    clk = np.median(b[:, :, 4] / np.maximum(b[:, :, 6], 1)) * 0.1
    tiles = np.median(b[:, :, 5]); nh = np.median(b[:, :, 7])
    fl = 2.0 * cin * cout * 9 * hw * hw * batch
    print(f"conv {cin}->{cout} @{hw}^2 x{batch}: {ms[1]/n:.3f} ms = {fl*4/9/(ms[1]/n*1e-3)/1e12:.1f} TFLOP/s executed; clock {clk:.3f} GHz; tiles/WG {tiles:.0f}; halves/tile {nh:.0f}; ideal loop cycles/tile {nh*4096:.0f}")
    for w in range(8):
        r = b[:, w, :]
        print(f"   wave {w} (xi {w&3} mh {w>>2}): per tile: loop {np.median(r[:,0]/r[:,5]):8.0f} ({np.median(r[:,0]/r[:,5])/nh:.0f}/half)  to-barrier1 {np.median(r[:,1]/r[:,5]):6.0f}  to-barrier2 {np.median(r[:,2]/r[:,5]):6.0f}  stores {np.median(r[:,3]/r[:,5]):6.0f}  | total/tile {np.median(r[:,4]/r[:,5]):8.0f}")


if __name__ == "__main__":
    run(64, 64, 256)
    run(256, 256, 64)
