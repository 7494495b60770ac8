"""Time the row-GEMM kernel variants (csrc/convnext_kernels.hip PH_GEMM_VARIANTS) on the shapes of cfg4.

    python tools/gemm_bench.py [batch]
"""
import ctypes as C
import sys

sys.path.insert(0, ".")
from sleap_nn_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = L.lib()
shapes = [
    # name, M, K, N, mode, H, W, act
    ("s0 lin 96->384 gelu", B * 192 * 192, 96, 384, 0, 0, 0, 2),
    ("s0 lin 384->96", B * 192 * 192, 384, 96, 0, 0, 0, 0),
    ("s1 lin 192->768 gelu", B * 96 * 96, 192, 768, 0, 0, 0, 2),
    ("s1 lin 768->192", B * 96 * 96, 768, 192, 0, 0, 0, 0),
    ("s2 lin 384->1536 gelu", B * 48 * 48, 384, 1536, 0, 0, 0, 2),
    ("s2 lin 1536->384", B * 48 * 48, 1536, 384, 0, 0, 0, 0),
    ("s3 lin 768->3072 gelu", B * 24 * 24, 768, 3072, 0, 0, 0, 2),
    ("s3 lin 3072->768", B * 24 * 24, 3072, 768, 0, 0, 0, 0),
    ("s0 lin 96->384 noact", B * 192 * 192, 96, 384, 0, 0, 0, 0),
    ("s2 lin 384->1536 noact", B * 48 * 48, 384, 1536, 0, 0, 0, 0),
    ("dec0 conv 2304->768 @24", B * 24 * 24, 2304, 768, 2, 24, 24, 1),
    ("dec1 conv 1152->384 @48", B * 48 * 48, 1152, 384, 2, 48, 48, 1),
    ("mid conv 1536->1536 @12", B * 12 * 12, 1536, 1536, 2, 12, 12, 1),
]
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4]
only = sys.argv[3] if len(sys.argv) > 3 else ""
shapes = [x for x in shapes if only in x[0]]
print("variant:           " + "".join(f"{v:>9d}" for v in variants))
for name, M, K, N, mode, H, W, act in shapes:
    taps = {0: 1, 1: 4, 2: 9}[mode]
    flops = 2.0 * M * K * taps * N
    row = []
    for v in variants:
        ms = C.c_float()
        rc = lib.ph_debug_gemm_bench(v, M, K, N, mode, H, W, act, 5, C.byref(ms))
        if rc != 0:
            row.append("     err")
            continue
        row.append(f"{flops / ms.value / 1e9:8.1f}")
    print(f"{name:26s} " + " ".join(row), flush=True)
