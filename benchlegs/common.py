"""Constants, synthetic inputs and the timing / FLOP-accounting helpers the bench legs share (see bench.py's docstring for the contract)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

__all__ = ['WARM_MS', 'CFG3_BB', 'NODES', 'CFG3_HEADS', 'SIZE', 'MFMA_F32_PEAK_TFLOPS', 'MFMA_F16_PEAK_TFLOPS', 'CFG4_BB', 'CFG4_HEADS', 'synthetic_instances', 'rendered_heads', 'percentiles', 'conv_kernel_short_names', 'conv_kernel_long_names', 'forward_executed_flops', 'ROOT']

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG3_BB = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 32, "stem_stride": None,
           "middle_block": True, "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 4}


NODES = [f"n{i}" for i in range(13)]


CFG3_HEADS = {"confmaps": {"part_names": NODES, "sigma": 2.5, "output_stride": 4, "loss_weight": 1.0},
              "pafs": {"edges": [[NODES[i], NODES[i + 1]] for i in range(12)], "sigma": 75.0, "output_stride": 8, "loss_weight": 1.0}}


SIZE = 1024


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = 64 FLOP/clk/SIMD * 1024 SIMDs * 2.4 GHz


MFMA_F16_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (no sparsity)


CFG4_BB = {"in_channels": 1, "model_type": "tiny", "arch": None, "stem_patch_kernel": 4, "stem_patch_stride": 2, "kernel_size": 3, "filters_rate": 2,
           "convs_per_block": 2, "up_interpolate": True, "output_stride": 2, "max_stride": 32}


CFG4_HEADS = {"confmaps": {"part_names": NODES, "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0, "anchor_part": None}}


def synthetic_instances(batch: int, distinct: int = 8, size: int = SIZE) -> torch.Tensor:
    """(B, 6, 13, 2) keypoints: per frame 6 animals, centres U(150, S-150), node offsets N(0, 40 px), seed 777+b
    (BASELINE.md section 3); `distinct` different frames, repeated to fill the batch."""
    pts = []
    for b in range(min(batch, distinct)):
        rng = np.random.RandomState(777 + b)
        centres = rng.uniform(min(150, size / 4), size - min(150, size / 4), size=(6, 1, 2))
        pts.append(np.clip(centres + rng.normal(0, 40, size=(6, 13, 2)), 8, size - 9).astype(np.float32))
    reps = (batch + len(pts) - 1) // len(pts)
    return torch.from_numpy(np.stack(pts)).repeat(reps, 1, 1, 1)[:batch].contiguous()


def rendered_heads(batch: int, device):
    """Rendered head outputs for the post-process stage, drawn by the product's own target renderers (the
    reference's generate_multiconfmaps / generate_pafs semantics): confmaps (B,13,256,256) with sigma 2.5 at
    stride 4, PAFs (B,24,128,128) with sigma 75 at stride 8, 6 instances per frame."""
    from sleap_nn_amd.data.targets import generate_multiconfmaps, generate_pafs

    pts = synthetic_instances(batch).to(device)
    edges = [(i, i + 1) for i in range(12)]
    cms = generate_multiconfmaps(pts, (SIZE, SIZE), sigma=2.5 * 4 / 2 / 4, output_stride=4)  # sigma * stride = 5 px
    pafs = generate_pafs(pts, (SIZE, SIZE), sigma=75.0, output_stride=8, edge_inds=edges)
    return cms, pafs


def percentiles(ms):
    a = np.asarray(ms, dtype=np.float64)
    return {"median": float(np.median(a)), "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90)), "n": int(a.size)}


def _cfg5_traffic():
    """(HBM bytes of one cfg5 forward, source file) from the newest profiles/*_f16_cfg5_traffic.json (tools/run_profile_f16_cfg5.sh), or (None, None)."""
    import glob

    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_f16_cfg5_traffic.json")))
    if not c:
        return None, None
    try:
        return float(json.load(open(c[-1]))["forward"]["hbm_bytes"]), os.path.relpath(c[-1], ROOT)
    except Exception:
        return None, None


def conv_kernel_short_names():
    """PH_KV_* code of a 3x3 conv launch -> the key its launches are summed under in `roofline.kernels` (tests/test_bench_line_cpu.py: every conv family the library can report has one)."""
    from sleap_nn_amd import _lib as L

    return {L.KV_DIRECT: "direct", L.KV_WINO1D: "wino1d", L.KV_WINO2D: "wino2d", L.KV_W16: "w16", L.KV_C16: "c16", L.KV_ROWGEMM: "rowgemm", L.KV_WINO4: "wino4", L.KV_F16: "f16",
            L.KV_WINO2D_KS: "wino2d", L.KV_SMALLMAP: "smallmap", L.KV_F16_ROWS: "f16rows", L.KV_F16_BLOCK: "f16block"}  # (the split-K launches of small batches are the same kernel family)


def conv_kernel_long_names(precision="exact"):
    return {"wino2d": "conv3x3_wino2d_kernel<64> (Winograd F(2x2,3x3), 4/9 of the direct MFMA work)",
            "wino4": "conv3x3_wino4_kernel (Winograd F(4x4,3x3), 1/4 of the direct MFMA work; the decoder's bilinear x2 folded into its input transform)",
            "w16": "conv3x3_w16_kernel<1|2> (wave-private Winograd F(2x2,3x3) on the 16x16x4 MFMA, Cout 32, 4/9 of the direct MFMA work)",
            "wino1d": "conv3x3_wino_persist_kernel<64|32> (Winograd F(2,3) along x, 2/3 of the direct MFMA work)",
            "direct": "conv3x3_mfma_dma_persist_kernel<64|32> (direct)", "c16": "conv3x3_c16_kernel (direct)", "rowgemm": "gemm_mfma_dma_kernel<2> (9-tap row GEMM, direct)",
            "smallmap": "conv3x3_sm_kernel (Winograd F(2x2,3x3) on 8x8-pixel x 16-channel units: small maps at small per-rank batches, 4/9 of the direct MFMA work)",
            "f16": f"conv3x3_f16_persist_kernel<64|32, {3 if precision == 'split' else 1}> (direct, fp16 matrix pipe)",
            "f16rows": "conv3x3_f16_rows_kernel (direct, plain fp16 on v_mfma_f32_16x16x32_f16: row tiles, loader waves, weights L2 -> registers, folded bilinear x2)",
            "f16block": "block2_c32_f16_kernel (the two convs of a 32-channel encoder block in one launch, plain fp16)"}


def _pad16(c):
    return (c + 15) // 16 * 16


def _matrix_rows(table):
    from sleap_nn_amd import _lib as L

    return [r for r in table if r["kind"] in (L.OP_CONV, L.OP_INPUT_CONV, L.OP_LINEAR, L.OP_PATCH_CONV, L.OP_PATCH_STEM)]


def forward_executed_flops(table, codes):
    """FLOPs the matrix cores execute in one forward, priced per launch by the kernel family the library reports it ran
    (ph_model_last_kernels): direct kernels and row GEMMs the direct count, F(2,3) 2/3, F(2x2,3x3) 4/9, F(4x4,3x3) 1/4.
    First convs on the VALU (input conv, patch stem) and the fused stem's VALU conv are not matrix work."""
    from sleap_nn_amd import _lib as L

    ex = 0.0
    for r, code in zip(table, codes):
        if r["kind"] == L.OP_STEM:
            ex += r["mfma_flops"] * L.KV_MFMA_SHARE[L.KV_STEM]
        elif r["kind"] in (L.OP_CONV, L.OP_LINEAR, L.OP_PATCH_CONV):
            ex += r["flops"] * L.KV_MFMA_SHARE.get(code, 1.0)
    return ex


def _forward_profile(model, x, n=10):
    """Per-op HIP-event pass of the eager forward: (op table, per-op ms, kernel codes, executed / direct matrix FLOPs, matrix ms)."""
    from sleap_nn_amd import _lib as L

    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    model.set_profiling(True)
    for _ in range(n):
        model(x)
    torch.cuda.synchronize()
    op_ms, n_fw = model.read_profile()
    model.set_profiling(False)
    codes = model.last_kernels()
    B, _, H, W = x.shape
    table = model.op_table(B, H, W)
    op_ms = [t / max(n_fw, 1) for t in op_ms]
    executed = forward_executed_flops(table, codes)
    direct = sum(r["flops"] for r in table)
    matrix_ms = sum(t for r, t in zip(table, op_ms) if r["kind"] in (L.OP_CONV, L.OP_STEM, L.OP_CONVT) and t > 0)
    kernels = {}
    for r, t, c in zip(table, op_ms, codes):
        if c != L.KV_NONE and c != L.KV_FUSED:
            e = kernels.setdefault(L.KV_NAMES[c].split(" (")[0], {"launches": 0, "ms": 0.0})
            e["launches"] += 1
            e["ms"] += t
    return table, op_ms, codes, executed, direct, matrix_ms, kernels


def _small_roofline(executed, direct, matrix_ms, fwd_s, kernels, n_ops):
    return {"bound": "mfma", "kernel": "whole conv stack of the forward (F(2x2,3x3) kernels incl. their split-K form, wave-private kernel, fused stem): at these sizes every layer has fewer work units than the "
                                       "256 CUs, the launches are latency-bound and the figure says how far from the matrix pipe that leaves them",
            "achieved": executed / fwd_s / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": executed / fwd_s / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "flop_accounting": "achieved = FLOPs the MFMA pipe executes in one forward (launches priced by the kernel family the library reports) / the forward's wall time (hipGraph replay, back to back); "
                               "direct_equivalent_tflops = direct-convolution FLOPs / the same time",
            "direct_equivalent_tflops": direct / fwd_s / 1e12, "executed_gflop_per_forward": executed / 1e9, "direct_gflop_per_forward": direct / 1e9,
            "matrix_launch_ms_per_forward_with_events": matrix_ms, "ops_per_forward": n_ops, "kernels": kernels, "traffic": None}


WARM_MS = 60.0  # untimed calls in front of a timed loop run at least this long: a leg starts after seconds of host-side set-up, i.e. on a GPU at idle clocks, and a handful of
                # sub-millisecond warm-up calls end before the clocks are back (the headline's first two 10.6-ms steps ran 1.5 ms slow each for the same reason: DESIGN section 5)


def _time_calls(fn, steps, warmup, sync_each):
    t_w, k = time.perf_counter(), 0
    while k < warmup or 1e3 * (time.perf_counter() - t_w) < WARM_MS:
        fn()
        k += 1
        if k % 16 == 0:
            torch.cuda.synchronize()  # (the wall clock above should see GPU time, not the depth of the launch queue)
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t = time.perf_counter()
        fn()
        if sync_each:
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ts
