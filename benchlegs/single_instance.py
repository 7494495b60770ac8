"""bench.py leg: BASELINE cfg1 / cfg2: single-instance UNets (legs `infer_cfg1`, `infer_cfg2`)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from benchlegs.common import *  # noqa: F401,F403  (constants + helpers; the names are listed in common.__all__)
from benchlegs.common import ROOT, _cfg5_traffic, _forward_profile, _matrix_rows, _pad16, _small_roofline, _time_calls  # noqa: F401


SI_BB = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 16, "stem_stride": None, "middle_block": True, "up_interpolate": True,
         "stacks": 1, "convs_per_block": 2, "output_stride": 2}


def single_instance_leg(name, size, batch, n_nodes, steps, dev, with_cpu):
    """BASELINE cfg1 / cfg2: single-instance UNet f16/r2/max_stride 16/output_stride 2.  A step = uint8 frames (resident in HBM) -> forward (one hipGraph replay) -> global peaks +
    integral refinement -> D2H of the keypoints.  cfg1 (one frame) is a latency workload: median / p90 of the synchronous per-frame time; cfg2 (8 frames) a throughput one."""
    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.inference.backends import HipBackend
    from sleap_nn_amd.inference.layers import PostprocessConfig, SingleInstanceLayer

    heads = {"confmaps": {"part_names": [f"k{i}" for i in range(n_nodes)], "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0}}
    model = Model("unet", SI_BB, heads, "single_instance").init_xavier_(seed=1234, head_scale=0.05).to(dev)
    g = torch.Generator().manual_seed(4321)
    frames = torch.randint(0, 256, (batch, 1, size, size), dtype=torch.uint8, generator=g).to(dev)
    table, op_ms, codes, executed, direct, matrix_ms, kernels = _forward_profile(model, frames)
    backend = HipBackend(model, str(dev), use_graph=True)
    layer = SingleInstanceLayer(backend, 2, max_stride=16, postprocess_config=PostprocessConfig(peak_threshold=0.0))
    # the frames live in the graph's own input buffer (HipBackend.static_input: where a pipeline's H2D copy would land them): a step is one graph launch, no staging copy
    frames = backend.static_input(tuple(frames.shape)).copy_(frames)
    fwd_total, _ = _time_calls(lambda: backend(frames), max(steps, 200), 20, False)
    fwd_s = fwd_total / max(steps, 200)
    # a step = InferenceLayer.predict_graphed: forward AND post-process (global peaks, refinement, coordinate ladder) captured as one graph, fed from the graph's own input buffer
    gframes = layer.graph_input(tuple(frames.shape)).copy_(frames)
    ref_out = layer.predict(frames)
    got_out = layer.predict_graphed(gframes)
    assert torch.equal(torch.nan_to_num(ref_out.pred_keypoints), torch.nan_to_num(got_out.pred_keypoints)) and torch.equal(ref_out.pred_peak_values, got_out.pred_peak_values)
    kp_host = torch.empty(tuple(got_out.pred_keypoints.shape), dtype=torch.float32, pin_memory=True)
    pv_host = torch.empty(tuple(got_out.pred_peak_values.shape), dtype=torch.float32, pin_memory=True)

    def latency_step():  # ends with the keypoints and their values in (pinned) host memory, as the CPU baseline beside it does
        o = layer.predict_graphed(gframes)
        kp_host.copy_(o.pred_keypoints, non_blocking=True)
        pv_host.copy_(o.pred_peak_values, non_blocking=True)

    total, lat = _time_calls(latency_step, steps, 10, True)
    lat_us = sorted(1e6 * t for t in lat)
    # throughput: the same steps enqueued back to back (the layer's outputs stay on the device: no host sync inside a step), one sync at the end
    total_q, _ = _time_calls(lambda: layer.predict_graphed(gframes), steps, 10, False)
    total_eager_q, _ = _time_calls(lambda: layer.predict(frames), steps, 10, False)
    two = None
    if batch > 1:  # a batch is a throughput workload (`value` = queued steps); one frame is a latency workload (`value` = 1 / median synchronous step)
        total = total_q
    else:
        total = steps * lat_us[len(lat_us) // 2] * 1e-6
    if True:
        # ... and the same queued steps alternating between TWO copies of the network on two HIP streams: most launches of a small step have fewer work units than CUs
        model2 = Model("unet", SI_BB, heads, "single_instance").init_xavier_(seed=1234, head_scale=0.05).to(dev)
        layer2 = SingleInstanceLayer(HipBackend(model2, str(dev), use_graph=True), 2, max_stride=16, postprocess_config=PostprocessConfig(peak_threshold=0.0))
        g2 = layer2.graph_input(tuple(frames.shape)).copy_(frames)
        assert torch.equal(torch.nan_to_num(layer2.predict_graphed(g2).pred_keypoints), torch.nan_to_num(got_out.pred_keypoints))
        from sleap_nn_amd.inference.predictor import concurrent_streams

        sts = concurrent_streams(dev, 2)
        pair = ((layer, gframes), (layer2, g2))
        cnt = [0]

        def step2():
            k = cnt[0] & 1
            cnt[0] += 1
            with torch.cuda.stream(sts[k]):
                pair[k][0].predict_graphed(pair[k][1])

        total_2, _ = _time_calls(step2, steps, 10, False)
        two = {"value": batch * steps / total_2, "unit": "frames/s", "what": "the queued steps alternating between two copies of the network on two HIP streams (outputs left on the device): a throughput figure"}
        del layer2, model2
    res = {"metric": f"frames/sec single-instance UNet {size}x{size} inference (batch {batch})", "value": batch * steps / total, "unit": "frames/s", "steps": steps, "ms_per_step": 1e3 * total / steps,
           "queued_steps_frames_per_s": batch * steps / total_q, "queued_steps_frames_per_s_two_launch_groups": batch * steps / total_eager_q,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{name}: single-instance UNet f16/r2/max_stride16/output_stride2, {size}x{size}x1 uint8 frames, {n_nodes} keypoints, batch {batch}", "frames_per_step": batch,
                      "weights": "xavier-uniform seed 1234, head x0.05", "params": model.num_parameters(), "step": "forward + global peaks + integral refinement + coordinate ladder as ONE hipGraph launch (InferenceLayer.predict_graphed); the synchronous (latency) steps end with the D2H of keypoints and values into pinned memory, the queued (throughput) steps leave them on the device; `value`: one frame = synchronous steps, a batch = steps queued back to back; queued_steps_frames_per_s_two_launch_groups = layer.predict (forward graph, then the post-process launches)",
                      "inputs": "uint8 frames resident in HBM"},
           "latency_us_per_step": {"median": lat_us[len(lat_us) // 2], "p10": lat_us[len(lat_us) // 10], "p90": lat_us[(9 * len(lat_us)) // 10]},
           "forward_only": {"us_per_batch": 1e6 * fwd_s, "frames_per_s": batch / fwd_s, "launch": "hipGraph replay, back to back, no host sync"},
           "roofline": _small_roofline(executed, direct, matrix_ms, fwd_s, kernels, len(table))}
    if two is not None:
        res["two_streams"] = two
    if with_cpu:  # cfg1 IS the reference-CPU-path configuration of BASELINE.json: the oracle on this box's host cores, same weights, same frame, parity beside it
        from oracle import cpu_ref as O

        sd = model.state_dict()
        img = frames[:1].cpu()
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        out = {}
        for th in sorted({1, min(avail, 8), min(avail, 16)}):
            torch.set_num_threads(th)
            with torch.inference_mode():
                O.model_forward(sd, SI_BB, heads, "single_instance", img)
                n, t0 = 0, time.perf_counter()
                while n < 3 or (time.perf_counter() - t0 < 2.5 and n < 200):
                    ref = O.model_forward(sd, SI_BB, heads, "single_instance", img)
                    rk, rv = O.single_instance_postprocess(ref["SingleInstanceConfmapsHead"], 2)
                    n += 1
                out[th] = (time.perf_counter() - t0) / n
        best = min(out, key=out.get)
        torch.set_num_threads(avail)
        got = model(frames[:1])["SingleInstanceConfmapsHead"].cpu()
        res["cpu_baseline"] = {"value": 1.0 / out[best], "unit": "frames/s", "cores": best, "kind": "port", "value_1thread": 1.0 / out[1], "ms_per_frame_by_threads": {str(k): 1e3 * v for k, v in out.items()},
                               "sample": "oracle/cpu_ref.py forward + global peaks of one 256x256 frame, ~2.5 s per thread count, torch-CPU fp32",
                               "parity_on_this_sample": {"max_abs_confmap_diff": float((got - ref["SingleInstanceConfmapsHead"]).abs().max()), "confmap_abs_max": float(ref["SingleInstanceConfmapsHead"].abs().max())}}
    return res
