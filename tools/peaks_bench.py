"""find_local_peaks on the cfg3 rendered maps (32 x 13 x 256 x 256): one-pass kernels vs the three-pass kernels, us per batch and GB/s of the maps read once."""
import ctypes as C
import sys

sys.path.insert(0, ".")
import torch

import bench
from sleap_nn_amd import _lib as L

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cms, _ = bench.rendered_heads(B, dev)
if "sparse" in sys.argv:  # blobs in one frame of eight only: is the cost of the windows that hold a blob per window or per launch?
    cms = cms.clone()
    cms[torch.arange(B, device=dev) % 8 != 0] = 0
if "zeros" in sys.argv:  # nothing above the threshold: the streaming phase alone
    cms = torch.zeros_like(cms)
Bc, Cc, H, W = cms.shape
cap = 4096
xy = torch.empty((cap, 2), device=dev); vals = torch.empty((cap,), device=dev)
sb = torch.empty((cap,), dtype=torch.int32, device=dev); sc = torch.empty((cap,), dtype=torch.int32, device=dev)
counts = torch.zeros((2 + 2 * Bc,), dtype=torch.int32, device=dev)
for name, ints in (("three-pass", 2 * Bc * H + 2), ("one-pass", int(L.lib().ph_local_peaks_scratch_bytes(Bc, Cc, H, W)) // 4)):
    scratch = torch.empty((ints,), dtype=torch.int32, device=dev)
    def call():
        L.check(L.lib().ph_local_peaks(C.c_void_p(cms.data_ptr()), Bc, Cc, H, W, 0.2, 1, 5, C.c_void_p(xy.data_ptr()), C.c_void_p(vals.data_ptr()), C.c_void_p(sb.data_ptr()), C.c_void_p(sc.data_ptr()),
                                       C.c_void_p(counts.data_ptr()), cap, 4.0, C.c_void_p(scratch.data_ptr()), scratch.numel() * 4, L.current_stream_ptr()))
    for _ in range(10):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 100
    print(f"{name}: {us:.1f} us / batch of {Bc}, {int(counts[0])} peaks, {cms.numel() * 4 / us / 1e3:.0f} GB/s of the maps read once ({cms.numel() * 4 / us / 1e3 / 8000:.2f} of 8 TB/s)")
