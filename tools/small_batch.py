"""Small-batch regime: per-op HIP-event times + eager / hipGraph wall time of the forward for

    cfg1   single-instance UNet f16/r2/ms16/os2, 256 x 256, K = 5, B = 1
    cfg2   same UNet, 512 x 512, K = 13, B = 8
    cfg3   bottom-up UNet, 1024 x 1024, B given (default 4: the 8-GPU strong-scaling shard)
    pub    the reference's fixture bottom-up run directory (tests/golden/ckpt_dirs), 320 x 560, B = 4
           (docs/guides/inference-performance.md:40-48: the only workload the reference publishes)

    python tools/small_batch.py cfg1|cfg2|cfg3|pub [B=..] [S=..] [opt=value ...]
"""
import os
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend

SI_BB = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 16, "stem_stride": None, "middle_block": True, "up_interpolate": True,
         "stacks": 1, "convs_per_block": 2, "output_stride": 2}


def build(cfg, dev):
    if cfg in ("cfg1", "cfg2"):
        K = 5 if cfg == "cfg1" else 13
        heads = {"confmaps": {"part_names": [str(i) for i in range(K)], "output_stride": 2}}
        m = Model("unet", SI_BB, heads, "single_instance").init_xavier_(seed=1234, head_scale=0.05)
        return m.to(dev), (1, 256, 256) if cfg == "cfg1" else (8, 512, 512)
    if cfg == "cfg3":
        return Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to(dev), (4, 1024, 1024)
    if cfg == "pub":
        from sleap_nn_amd.inference.loaders import load_model_assets

        a = load_model_assets(os.path.join("tests", "golden", "ckpt_dirs", "minimal_instance_bottomup"))
        return a.build_model().to(dev), (4, 320, 560)
    raise SystemExit(cfg)


def main():
    cfg = sys.argv[1]
    dev = torch.device("cuda", 0)
    m, (B, H, W) = build(cfg, dev)
    opts = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        if k == "B":
            B = int(v)
        elif k == "S":
            H = W = int(v)
        else:
            opts[k] = int(v)
    for k, v in opts.items():
        m.set_option(k, v)
    x = torch.randint(0, 256, (B, 1, H, W), dtype=torch.uint8, device=dev)
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    m.set_profiling(True)
    N = 20
    for _ in range(N):
        m(x)
    torch.cuda.synchronize()
    ms, n = m.read_profile()
    m.set_profiling(False)
    codes = m.last_kernels()
    tab = m.op_table(B, H, W)
    tot = 0.0
    gflop = sum(r["flops"] for r in tab) / 1e9
    for r, t, c in zip(tab, ms, codes):
        t /= n
        tot += t
        if t > 0:
            sh = L.KV_MFMA_SHARE.get(c, 0)
            tf = r["flops"] * sh / t / 1e9 if t else 0
            print(f"{r['label']:44s} {L.KV_NAMES.get(c, '-')[:22]:22s} {str(r.get('out_hw')):12s} {t*1e3:8.1f} us  exec {tf:6.1f} TF/s  direct {r['flops']/t/1e9:6.1f}")
    print(f"{cfg} options {opts} B={B} {H}x{W}: per-op sum {tot*1e3:.1f} us, {gflop:.2f} GFLOP direct / batch")
    for graph in (False, True):
        be = HipBackend(m, "cuda:0", use_graph=graph)
        xx = x.unsqueeze(1)
        for _ in range(10):
            be(xx)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t = time.perf_counter()
            for _ in range(50):
                be(xx)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) / 50)
        dt = sorted(ts)[len(ts) // 2]
        print(f"  graph={graph}: {dt*1e6:.1f} us / batch = {B/dt:.0f} frames/s, {gflop/dt/1e3:.1f} TFLOP/s direct-equivalent")


main()
