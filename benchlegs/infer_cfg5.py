"""bench.py leg: BASELINE cfg5: multi-class bottom-up on the fp16 matrix pipe (leg `infer_cfg5`)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from benchlegs.common import *  # noqa: F401,F403  (constants + helpers; the names are listed in common.__all__)
from benchlegs.common import ROOT, _cfg5_traffic, _forward_profile, _matrix_rows, _pad16, _small_roofline, _time_calls  # noqa: F401


def infer_cfg5_leg(steps, dev):
    """BASELINE cfg5: multi-class bottom-up, 768 x 768, 4 classes x 17 keypoints, fp16 MFMA, batch 16.  The reference has no HRNet: the backbone is its UNet
    (the cfg3 architecture) with a class-maps head; the forward runs in the autocast-equivalent fp16 precision (fp16 storage and MFMA operands, fp32 accumulation and head outputs,
    torch_backend.py:113-143).  A step = forward (hipGraph replay) + local peaks + class-map sampling + D2H + host grouping by class, on rendered heads (one animal per class), synchronous."""
    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.data.targets import generate_multiconfmaps
    from sleap_nn_amd.inference.backends import HipBackend
    from sleap_nn_amd.inference.layers import BottomUpMultiClassLayer, PostprocessConfig
    from sleap_nn_amd.inference.preprocess_info import PreprocInfo

    S, N, K, B = 768, 17, 4, 16
    heads = {"confmaps": {"part_names": [f"k{i}" for i in range(N)], "sigma": 2.5, "output_stride": 4, "loss_weight": 1.0},
             "class_maps": {"classes": [f"id{i}" for i in range(K)], "sigma": 12.5, "output_stride": 8, "loss_weight": 1.0}}
    model = Model("unet", dict(CFG3_BB), heads, "multi_class_bottomup").init_xavier_(seed=1234, head_scale=0.05).to(dev)
    g = torch.Generator().manual_seed(4321)
    frames = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, generator=g).to(dev)
    exact = {k: v.clone() for k, v in model(frames[:2]).items()}
    backend = HipBackend(model, str(dev), use_graph=True, use_fp16=True)
    layer = BottomUpMultiClassLayer(backend, 4, 8, max_stride=32, postprocess_config=PostprocessConfig(peak_threshold=0.2))
    got = backend(frames[:2].contiguous())
    drift = {k: float((got[k] - exact[k]).abs().max()) for k in exact}
    # rendered heads: one animal per class, 17 nodes each; class maps = blobs around that animal's nodes
    rng = np.random.RandomState(3)
    pts = np.stack([np.clip(rng.uniform(120, S - 120, size=(K, 1, 2)) + rng.normal(0, 35, size=(K, N, 2)), 6, S - 7) for _ in range(B)]).astype(np.float32)
    cms = generate_multiconfmaps(torch.from_numpy(pts).to(dev), (S, S), sigma=2.5 * 4 / 2 / 4, output_stride=4)  # as rendered_heads: sigma * stride = 5 px
    yy, xx = torch.meshgrid(torch.arange(0, S, 8, dtype=torch.float32, device=dev), torch.arange(0, S, 8, dtype=torch.float32, device=dev), indexing="ij")
    tp = torch.from_numpy(pts).to(dev)
    d2 = (xx[None, None, None] - tp[..., 0, None, None]) ** 2 + (yy[None, None, None] - tp[..., 1, None, None]) ** 2
    cmaps = torch.exp(-d2 / (2 * 50.0**2)).amax(2)
    info = PreprocInfo(eff_scale=torch.ones(B))
    fb = backend.static_input(tuple(frames.shape)).copy_(frames)

    def step():
        backend(fb)
        return layer.postprocess({"MultiInstanceConfmapsHead": cms, "ClassMapsHead": cmaps}, info)

    out = step()
    found = int((~torch.isnan(out.pred_keypoints[..., 0])).sum())
    n = max(steps, 50)
    fwd_total, _ = _time_calls(lambda: backend(fb), n, 10, False)
    total_sync, lat = _time_calls(step, n, 5, True)
    # pipelined, as the bottom-up predictor runs its batches: the GPU stage of step i + 1 (forward + peaks + class-map sampling + async D2H) is enqueued before the host
    # stage of step i (Hungarian matching by class in a worker thread) is collected
    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(max_workers=1)
    futs = []

    def pstep():
        backend(fb)
        futs.append(pool.submit(layer._finish_postprocess, layer._enqueue_postprocess({"MultiInstanceConfmapsHead": cms, "ClassMapsHead": cmaps}, info)))
        if len(futs) > 2:
            futs.pop(0).result()

    t_w, k = time.perf_counter(), 0
    while k < 5 or 1e3 * (time.perf_counter() - t_w) < WARM_MS:  # (untimed: the timed loop below must not start at the clocks of a GPU that idled through the set-up above)
        pstep()
        k += 1
    last = [f.result() for f in futs][-1]
    futs.clear()
    assert torch.equal(torch.nan_to_num(last.pred_keypoints), torch.nan_to_num(out.pred_keypoints))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        pstep()
    for f in futs:
        f.result()
    torch.cuda.synchronize()
    total = time.perf_counter() - t1
    futs.clear()
    pool.shutdown()
    fwd_s = fwd_total / n
    table = model.op_table(B, S, S)
    direct = sum(r["flops"] for r in table)
    return {"metric": "frames/sec multi-class bottom-up UNet 768x768 inference, fp16 MFMA (batch 16)", "value": B * n / total, "unit": "frames/s", "steps": n, "ms_per_step": 1e3 * total / n,
            "dtype": "f16 (f32 accumulate)", "data": "synthetic",
            "config": {"workload": "cfg5: multi-class bottom-up, UNet f16/r2/max_stride32/output_stride4 backbone (the reference has no HRNet), 768x768x1 uint8 frames, 4 classes x 17 keypoints, batch 16",
                       "frames_per_step": B, "params": model.num_parameters(), "postprocess_input": "rendered heads, one animal per class", "keypoints_found_per_step": found,
                       "step": "forward (hipGraph replay, fp16 pipe) + local peaks + class-map sampling + async D2H, host grouping by class in a worker thread; steps pipelined (the next GPU stage is enqueued before this step's host stage is collected), every step grouped before the clock stops"},
            "synchronous_steps": {"value": B * n / total_sync, "unit": "frames/s", "ms_per_step": 1e3 * total_sync / n, "what": "the same step with a host sync behind each (round 4's definition of this leg)"},
            "forward_only": {"ms_per_batch": 1e3 * fwd_s, "frames_per_s": B / fwd_s},
            "max_abs_head_diff_vs_exact_fp32": drift, "head_abs_max": {k: float(v.abs().max()) for k, v in exact.items()},
            "roofline": {"bound": "mfma", "kernel": "stem_f16_kernel + block2_c32_f16_kernel + conv3x3_f16_rows_kernel / conv3x3_f16_persist_kernel (direct 3x3 on v_mfma_f32_16x16x32_f16 / 32x32x16_f16; bilinear x2 and both heads folded) over the whole forward", "achieved": direct / fwd_s / 1e12, "peak": MFMA_F16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": direct / fwd_s / 1e12 / MFMA_F16_PEAK_TFLOPS, "direct_gflop_per_forward": direct / 1e9,
                         "flop_accounting": "direct-convolution FLOPs of the forward (this pipe runs the direct form: executed = direct) / the forward's wall time (hipGraph replay, back to back), against the dense fp16 MFMA peak",
                         "traffic": _cfg5_traffic()[0], "traffic_source": _cfg5_traffic()[1], "traffic_unit": "HBM bytes per forward (PMC: FETCH_SIZE x 2 + WRITE_SIZE, separate passes)",
                         "algorithmic_bytes_per_forward": sum(r["bytes"] for r in table)}}
