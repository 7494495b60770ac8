"""bench.py leg: the training step of the cfg3 UNet / the cfg4 ConvNeXt-tiny (legs `train_cfg3`, `train_cfg4`; `--mode train`)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from benchlegs.common import *  # noqa: F401,F403  (constants + helpers; the names are listed in common.__all__)
from benchlegs.common import ROOT, _cfg5_traffic, _forward_profile, _matrix_rows, _pad16, _small_roofline, _time_calls  # noqa: F401


def train_leg(cfg, B, global_batch, steps, warmup, ctx, scaling="weak"):
    """Data-parallel training steps of one configuration: forward (unfused fp32 program) + per-head MSE + backward + two-bucket
    gradient all-reduce (RCCL, overlapped with the backward; nothing to reduce at N = 1) + Adam + re-pack of the kernel weights.
    ``cfg`` "cfg3": the bottom-up UNet at 1024x1024; "cfg4": BASELINE cfg4, ConvNeXt-tiny centered-instance on 384x384 crops.
    Returns the leg's dict on rank 0 (None elsewhere); also used as the headline of ``--mode train``."""
    rank, world, dev, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["dist"]
    from sleap_nn_amd import _lib as L
    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.data.targets import generate_multiconfmaps, generate_pafs
    from sleap_nn_amd.training.module import TrainingModule

    cfg4 = cfg == "cfg4"
    size = 384 if cfg4 else SIZE
    model = Model("convnext", CFG4_BB, CFG4_HEADS, "centered_instance") if cfg4 else Model("unet", CFG3_BB, CFG3_HEADS, "bottomup")
    model.init_xavier_(seed=1234, head_scale=0.05)
    tm = TrainingModule(model, str(dev), lr=1e-4)
    g = torch.Generator().manual_seed(4321 + rank)
    frames = torch.randint(0, 256, (B, 1, size, size), dtype=torch.uint8, generator=g).to(dev)
    pts = synthetic_instances(B, size=size).to(dev)
    if cfg4:
        targets = {"CenteredInstanceConfmapsHead": generate_multiconfmaps(pts[:, :1], (size, size), sigma=2.5 * 2 / 2 / 2, output_stride=2)}
    else:
        targets = {"MultiInstanceConfmapsHead": generate_multiconfmaps(pts, (SIZE, SIZE), sigma=2.5 * 4 / 2 / 4, output_stride=4),
                   "PartAffinityFieldsHead": generate_pafs(pts, (SIZE, SIZE), sigma=75.0, output_stride=8, edge_inds=[(i, i + 1) for i in range(12)])}
    batch = {"image": frames, **targets}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(max(warmup, 2)):
        losses.append(tm.training_step(batch).clone())
    codes = model.last_kernels()  # kernels of the training program's forward (the unfused program)
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        losses.append(tm.training_step(batch).clone())
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    first, last = float(losses[0][0]), float(losses[-1][0])
    assert np.isfinite(last) and last <= first, f"training loss did not go down: {first} -> {last}"
    # the gradient exchange alone (both buckets, nothing to overlap with), for scale
    ar_ms = None
    if world > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            tm.all_reduce_grads()
        barrier()
        e0.record()
        for _ in range(10):
            tm.all_reduce_grads()
        e1.record()
        barrier()
        ar_ms = e0.elapsed_time(e1) / 10
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    arena_mb, split = tm.grads.numel() * 4 / 1e6, tm._bucket_split
    n_params = model.num_parameters()
    table = model.op_table(B, size, size)
    tm.close()
    del tm, batch, targets, frames
    if rank != 0:
        return None, model
    fwd_flops = sum(r["flops"] for r in _matrix_rows(table))
    # Executed on the matrix pipe per step.  Forward: per launch by the kernel the library reports (ph_model_last_kernels).  Backward:
    # the data gradient of a 3x3 conv is the same kernel family on swapped channel counts (one conv per concat source) -- F(2x2,3x3)
    # (4/9) for N tiles of 64 output channels with >= 32 input channels (conv3x3_wino2d_kernel) and for one-source layers with 16 / 32
    # channels on both sides (conv3x3_w16_kernel), F(4x4,3x3) (1/4) from 128 input channels on, F(2,3) along x (2/3) otherwise -- and every 3x3 weight gradient runs in the
    # F(2x2,3x3) domain (4/9: wgrad_wino_kernel, wgrad16_wino_kernel); row GEMMs (Linear, 2x2/s2 convs) run direct in all three.
    def share(cin_p, cout_p, one_source=True, hw=(0, 0)):
        h, w = hw
        if cout_p >= 64 and cin_p >= 128 and h > 0 and h % 4 == 0 and w % 4 == 0:
            # conv3x3_wino4_kernel (TrainingModule runs it in the training plan: conv_wino4 = 2) where wino4_fits estimates it faster: rounds of the chip x time per tile
            ntc, n_cu = -(-cout_p // 64), 256
            t4, t2 = -(-h // 16) * -(-w // 32) * B * ntc, -(-h // 16) * -(-w // 16) * B * ntc
            if -(-t4 // n_cu) * (2.0 / 1.3) <= -(-t2 // n_cu):
                return 0.25
        if cout_p >= 64 and cin_p >= 32:
            return 4.0 / 9.0
        if one_source and cout_p in (16, 32) and cin_p in (16, 32):
            return 4.0 / 9.0
        return 2.0 / 3.0
    executed = forward_executed_flops(table, codes)
    for r in table:
        if r["kind"] in (L.OP_LINEAR, L.OP_PATCH_CONV):
            executed += 2.0 * r["flops"]
        elif r["kind"] == L.OP_CONV and r.get("ksize", 3) == 3:
            cin = r["cin0"] + r["cin1"]
            for part in (r["cin0"], r["cin1"]):  # data gradient: one conv per concat source, Cout -> part channels
                if part > 0:
                    executed += r["flops"] * part / cin * share(_pad16(r["cout"]), _pad16(part), True, r.get("out_hw", (0, 0)))
            executed += r["flops"] * 4.0 / 9.0  # weight gradient
        elif r["kind"] in (L.OP_INPUT_CONV, L.OP_PATCH_STEM):
            executed += r["flops"]  # weight gradient only (no data gradient into the image)
    per_step = elapsed / steps
    res = {
        "metric": "frames/sec training step (forward + MSE + backward + gradient all-reduce + Adam)",
        "value": global_batch * steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("cfg4: ConvNeXt-tiny centered-instance, 384x384 crops, 13 nodes, output stride 2, global batch 64" if cfg4 else
                                "cfg3 network in training: bottom-up UNet f16/r2/max_stride32/output_stride4, 1024x1024x1 frames, 13 nodes / 12 edges"),
                   "samples_per_gpu_per_step": B, "global_batch": global_batch, "parallelism": f"dp{world}: replicas, disjoint shards, two-bucket RCCL all-reduce overlapped with the backward",
                   "params": n_params, "optimizer": "Adam lr 1e-4", "targets": "rendered on the device by ph_render_confmaps / ph_render_pafs"},
        "step_ms": percentiles(step_ms),
        "loss_first_last": [first, last],
        "allreduce": {"arena_mb": arena_mb, "bucket_split": split, "standalone_ms": ar_ms,
                      "note": "standalone_ms = both buckets back to back with nothing to overlap (null at N = 1); in a step the tail bucket runs under the encoder's backward"},
        "roofline": {"bound": "mfma", "kernel": "forward + data-gradient convolutions and 3x3 weight gradients (Winograd F(2x2,3x3) / F(2,3) kernels) and row GEMMs, on v_mfma_f32_32x32x2_f32 / 16x16x4_f32",
                     "achieved": executed / per_step / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": executed / per_step / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     "flop_accounting": "whole step time in the denominator (loss, masks, pools, LayerNorm / GELU / depthwise, Adam, re-pack included); executed FLOPs = forward launches priced by the kernel family the library reports (ph_model_last_kernels) + data gradients of the same families + 3x3 weight gradients at 4/9 (Winograd domain) + row-GEMM gradients direct; direct_equivalent_tflops = 3 x forward matrix FLOPs / step time, a throughput figure, not a roofline fraction",
                     "executed_gflop_per_step": executed / 1e9,
                     "direct_equivalent_tflops": 3.0 * fwd_flops / per_step / 1e12,
                     "forward_matrix_gflop_per_step": fwd_flops / 1e9},
    }
    return res, model
