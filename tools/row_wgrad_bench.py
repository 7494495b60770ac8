"""Time the row weight-gradient GEMM (csrc/convnext_train_kernels.hip row_wgrad_kernel) on the Linear shapes of cfg4 and spot-check it
against a float64 host sum.    python tools/row_wgrad_bench.py [batch]"""
import ctypes as C
import sys

sys.path.insert(0, ".")
from sleap_nn_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = L.lib()
shapes = [("odd rows", 10007, 96, 48), ("s0 fc1 384x96", B * 192 * 192, 384, 96), ("s0 fc2 96x384", B * 192 * 192, 96, 384), ("s1 fc1 768x192", B * 96 * 96, 768, 192),
          ("s2 fc1 1536x384", B * 48 * 48, 1536, 384), ("s2 fc2 384x1536", B * 48 * 48, 384, 1536), ("s3 fc1 3072x768", B * 24 * 24, 3072, 768)]
for name, M, n, k in shapes:
    ms, err = C.c_float(), C.c_float()
    L.check(lib.ph_debug_row_wgrad_bench(M, n, k, 5, C.byref(ms), C.byref(err)))
    print(f"{name:18s} M {M:8d}: {ms.value:7.3f} ms  {2.0 * M * n * k / ms.value / 1e9:6.1f} TFLOP/s   max err / scale {err.value:.2e}", flush=True)
