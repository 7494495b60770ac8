"""bench.py leg: the cfg4 network's inference forward (leg `infer_cfg4`)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from benchlegs.common import *  # noqa: F401,F403  (constants + helpers; the names are listed in common.__all__)
from benchlegs.common import ROOT, _cfg5_traffic, _forward_profile, _matrix_rows, _pad16, _small_roofline, _time_calls  # noqa: F401


def infer_cfg4_leg(model, B, steps, warmup, dev):
    """Inference forward of the cfg4 network (ConvNeXt-tiny centered-instance, 384x384 crops, output stride 2) on B crops: the
    fused inference program (LayerNorms inside the depthwise / stem kernels), kernel by kernel (one event per timed step; per-op HIP events in a separate untimed pass)."""
    from sleap_nn_amd import _lib as L

    size = 384
    model.bind_live_params(None)
    model.eval().to(dev)
    g = torch.Generator().manual_seed(4321)
    crops = torch.randint(0, 256, (B, 1, size, size), dtype=torch.uint8, generator=g).to(dev)
    for _ in range(max(warmup, 2)):
        out = model(crops)
    codes = model.last_kernels()
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in out.values())
    # timed steps: kernel by kernel, one event per step; the per-op events (two per op, ~100 ops) ride in a second, untimed pass
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        model(crops)
        marks[i + 1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    model.set_profiling(True)
    for _ in range(max(steps // 2, 3)):
        model(crops)
    torch.cuda.synchronize()
    op_ms, n_fw = model.read_profile()
    model.set_profiling(False)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    table = model.op_table(B, size, size)
    mm = _matrix_rows(table)
    fwd_flops = sum(r["flops"] for r in mm)
    executed = forward_executed_flops(table, codes)
    per_step = elapsed / steps
    groups = {}
    NAMES = {L.OP_CONV: "conv3x3", L.OP_LINEAR: "linear (CNBlock MLP)", L.OP_PATCH_CONV: "conv2x2/s2", L.OP_PATCH_STEM: "patch stem (+LayerNorm2d)", L.OP_DWCONV: "depthwise 7x7 (+LayerNorm)",
             L.OP_LAYERNORM: "layernorm", L.OP_UPSAMPLE: "bilinear x2", L.OP_POOL: "pool", L.OP_HEAD: "head"}
    for r, ms, code in zip(table, op_ms, codes):
        e = groups.setdefault(NAMES.get(r["kind"], str(r["kind"])), {"launches": 0, "ms": 0.0, "direct_gflop": 0.0, "executed_gflop": 0.0})
        e["launches"] += 1 if ms > 0 else 0
        e["ms"] += ms / max(n_fw, 1)
        if r["kind"] in (L.OP_CONV, L.OP_LINEAR, L.OP_PATCH_CONV):
            e["direct_gflop"] += r["flops"] / 1e9
            e["executed_gflop"] += r["flops"] * L.KV_MFMA_SHARE.get(code, 1.0) / 1e9
    for e in groups.values():
        e["executed_tflops"] = e["executed_gflop"] / e["ms"] if e["ms"] > 0 else 0.0
    matrix_ms = sum(e["ms"] for e in groups.values() if e["executed_gflop"] > 0)
    return {
        "metric": "crops/sec ConvNeXt-tiny centered-instance inference forward", "value": B * steps / elapsed, "unit": "crops/s", "steps": steps, "ms_per_step": 1e3 * per_step,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg4 network at inference: ConvNeXt-tiny centered-instance, 384x384x1 uint8 crops, 13 nodes, output stride 2", "crops_per_step": B,
                   "params": model.num_parameters(), "forward_launch": "kernel by kernel; per-op HIP events in a separate untimed pass"},
        "step_ms": percentiles(step_ms),
        "roofline": {"bound": "mfma", "kernel": "row GEMMs (CNBlock MLPs, 2x2/s2 convs: gemm_mfma_dma_kernel) + decoder / middle 3x3 convs (F(2x2,3x3) and 9-tap row-GEMM forms), v_mfma_f32_32x32x2_f32",
                     "achieved": executed / (matrix_ms * 1e-3) / 1e12 if matrix_ms > 0 else 0.0, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": executed / (matrix_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if matrix_ms > 0 else 0.0,
                     "flop_accounting": "achieved = FLOPs the MFMA pipe executes in the matrix launches (3x3 convs priced by the kernel family the library reports, row GEMMs direct) / their summed duration (per-op HIP events of the untimed profiling pass); whole_forward_frac divides by the whole forward (depthwise, LayerNorm, bilinear, head included)",
                     "whole_forward_frac": executed / per_step / 1e12 / MFMA_F32_PEAK_TFLOPS,
                     "direct_equivalent_tflops": fwd_flops / per_step / 1e12, "executed_gflop_per_forward": executed / 1e9, "direct_gflop_per_forward": fwd_flops / 1e9,
                     "matrix_ms_per_forward": matrix_ms, "forward_ms": sum(op_ms) / max(n_fw, 1), "by_op_kind": groups},
    }
