"""Benchmark of the hot path: bottom-up UNet 1024x1024 inference (BASELINE.json cfg3).

    python bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one per-GPU batch of synthetic frames that are
already resident in HBM: uint8 frames -> UNet forward (random-init weights of the cfg3
architecture, every kernel a hand-written gfx950 kernel) -> local peaks + integral refinement ->
PAF line scoring -> D2H of the scored candidates -> C++ matching/assembly -> NaN-padded
keypoints on the host.  Because random weights give meaningless peaks, the post-process stage
consumes *rendered* heads with 6 animals/frame that are equally resident in
HBM; forward and post-process both run in full inside every step, back to back on one stream.

Multi-GPU: one process per GPU (torch.distributed / RCCL), frames sharded across ranks, no
data-path collective (frames are independent) -> weak scaling, value = total frames / max time.

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (dominant kernel =
the MFMA conv3x3, HIP-event timed inside the timed region) and `cpu_baseline` (the oracle on
the host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

CFG3_BB = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 32, "stem_stride": None,
           "middle_block": True, "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 4}
NODES = [f"n{i}" for i in range(13)]
CFG3_HEADS = {"confmaps": {"part_names": NODES, "sigma": 2.5, "output_stride": 4, "loss_weight": 1.0},
              "pafs": {"edges": [[NODES[i], NODES[i + 1]] for i in range(12)], "sigma": 75.0, "output_stride": 8, "loss_weight": 1.0}}
SIZE = 1024
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = 64 FLOP/clk/SIMD * 1024 SIMDs * 2.4 GHz


def synthetic_instances(batch: int, distinct: int = 8) -> torch.Tensor:
    """(B, 6, 13, 2) keypoints: per frame 6 animals, centres U(150, S-150), node offsets N(0, 40 px), seed 777+b
    (BASELINE.md section 3); `distinct` different frames, repeated to fill the batch."""
    pts = []
    for b in range(min(batch, distinct)):
        rng = np.random.RandomState(777 + b)
        centres = rng.uniform(150, SIZE - 150, size=(6, 1, 2))
        pts.append(np.clip(centres + rng.normal(0, 40, size=(6, 13, 2)), 8, SIZE - 9).astype(np.float32))
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


def cpu_baseline(sd, cms, pafs, budget_s: float = 20.0):
    """Oracle (torch-CPU restatement of the reference, kind = "port") on the host cores, on the same weights
    and the same rendered heads as the GPU leg.  The only place bench.py touches oracle/."""
    from oracle import cpu_ref as O

    g = torch.Generator().manual_seed(4321)
    img = torch.randint(0, 256, (1, 1, SIZE, SIZE), dtype=torch.uint8, generator=g)
    # pick the fastest thread count among a few candidates (oneDNN degrades badly when
    # oversubscribed on big hosts): one probe forward each, bounded
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best_t, best = avail, float("inf")
    with torch.inference_mode():
        for th in sorted({min(avail, c) for c in (avail, 64, 32, 16, 8)}, reverse=True):
            torch.set_num_threads(th)
            O.model_forward(sd, CFG3_BB, CFG3_HEADS, "bottomup", img[:, :, :512, :512])
            t = time.perf_counter()
            O.model_forward(sd, CFG3_BB, CFG3_HEADS, "bottomup", img[:, :, :512, :512])
            dt = time.perf_counter() - t
            if dt < best:
                best, best_t = dt, th
    torch.set_num_threads(best_t)
    scorer = O.PAFScorerRef(NODES, [tuple(e) for e in CFG3_HEADS["pafs"]["edges"]], 8)
    with torch.inference_mode():
        O.model_forward(sd, CFG3_BB, CFG3_HEADS, "bottomup", img)  # warm-up
        O.bottomup_postprocess(cms[:1], pafs[:1], scorer, 4)
        t_f, n_f = 0.0, 0
        t0 = time.perf_counter()
        while n_f < 3 or (time.perf_counter() - t0 < budget_s * 0.7 and n_f < 20):
            t = time.perf_counter()
            O.model_forward(sd, CFG3_BB, CFG3_HEADS, "bottomup", img)
            t_f += time.perf_counter() - t
            n_f += 1
        t_p, n_p = 0.0, 0
        t0 = time.perf_counter()
        while n_p < 2 or (time.perf_counter() - t0 < budget_s * 0.3 and n_p < 20):
            t = time.perf_counter()
            O.bottomup_postprocess(cms, pafs, scorer, 4)
            t_p += time.perf_counter() - t
            n_p += 1
    per_frame = t_f / n_f + t_p / (n_p * cms.shape[0])
    return {
        "value": 1.0 / per_frame, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{n_f} forwards of 1 frame 1024x1024 ({1e3 * t_f / n_f:.0f} ms each) + {n_p} post-process passes over 2 rendered frames "
                  f"({1e3 * t_p / (n_p * cms.shape[0]):.1f} ms/frame); oracle/cpu_ref.py, torch-CPU fp32",
    }


PROFILE_EVERY = 4  # steps of the timed region whose forward records per-op HIP events: 0, 4, 8, ...


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)
        # ranks share the host: keep torch's CPU pool (synthetic-input rendering only) to a fair share
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))

    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.inference.backends import HipBackend
    from sleap_nn_amd.inference.layers import BottomUpLayer
    from sleap_nn_amd.inference.ops.paf import PAFScorer
    from sleap_nn_amd.inference.preprocess_info import PreprocInfo

    B = args.batch
    model = Model("unet", CFG3_BB, CFG3_HEADS, "bottomup")
    model.init_xavier_(seed=1234, head_scale=0.05)  # the reference's xavier_init_weights; heads x0.05 keep outputs O(1)
    layer = BottomUpLayer(HipBackend(model, str(dev)), PAFScorer.from_config(CFG3_HEADS), 4, 8, max_stride=32)
    g = torch.Generator().manual_seed(4321 + rank)
    frames = torch.randint(0, 256, (B, 1, 1, SIZE, SIZE), dtype=torch.uint8, generator=g).to(dev)
    cms, pafs = rendered_heads(B, dev)
    info = PreprocInfo(original_size=(SIZE, SIZE), processed_size=(SIZE, SIZE), eff_scale=torch.ones(B), output_stride=4)

    from concurrent.futures import ThreadPoolExecutor

    from sleap_nn_amd.inference.streaming import group_scored_batch

    pool = ThreadPoolExecutor(max_workers=1)
    params = layer.grouping_params()
    pending = []

    inflight = []

    def step():
        """Forward + peaks + PAF scoring + async D2H are enqueued for this batch; then the PREVIOUS batch's
        results (its D2H event fired long ago) are handed to the C++ grouping worker.  The GPU always has the
        next batch queued and the grouping of batch k-1 overlaps the GPU work of batch k
        (Predictor._predict_streaming_pipelined).  Every batch is grouped before the closing barrier."""
        raw = layer.backend(frames)  # uint8 frames -> {"MultiInstanceConfmapsHead", "PartAffinityFieldsHead"}
        inflight.append(layer._enqueue_scoring({"MultiInstanceConfmapsHead": cms, "PartAffinityFieldsHead": pafs}, info))
        if len(inflight) > 1:
            pending.append(pool.submit(group_scored_batch, layer._finish_scoring(inflight.pop(0)), params))
        out = pending.pop(0).result() if len(pending) > 1 else None
        return raw, out

    def drain():
        while inflight:
            pending.append(pool.submit(group_scored_batch, layer._finish_scoring(inflight.pop(0)), params))
        outs = [f.result() for f in pending]
        pending.clear()
        return outs

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        raw, out = step()
    out = drain()[-1]
    torch.cuda.synchronize()
    n_inst = int((~torch.isnan(out.instance_scores)).sum())
    assert n_inst >= 5 * B, f"post-process found only {n_inst} instances in {B} frames"
    assert all(torch.isfinite(v).all() for v in raw.values())

    # HIP-event timing of every op runs INSIDE the timed region, on every PROFILE_EVERY-th step (26 event records per forward
    # cost ~0.6 ms of a 19 ms step; sampling a quarter of the steps keeps the roofline figures live at a quarter of that)
    model.set_profiling(True)
    model.set_profiling(False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.set_profiling(i % PROFILE_EVERY == 0, resume=True)
        step()
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    op_ms, n_fw = model.read_profile()
    model.set_profiling(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        from sleap_nn_amd import _lib as L

        table = model.op_table(B, SIZE, SIZE)
        conv_rows = [(r, ms / max(n_fw, 1)) for r, ms in zip(table, op_ms) if r["kind"] == L.OP_CONV]
        conv_flops = sum(r["flops"] for r, _ in conv_rows)
        conv_ms = sum(ms for _, ms in conv_rows)
        # whole conv stack = those launches + the fused stem (conv0 on VALU + conv1 on MFMA 16x16x4 + pool)
        stack_rows = [(r, ms / max(n_fw, 1)) for r, ms in zip(table, op_ms) if r["kind"] in (L.OP_CONV, L.OP_STEM, L.OP_INPUT_CONV)]
        stack_flops = sum(r["flops"] for r, _ in stack_rows)
        stack_ms = sum(ms for _, ms in stack_rows)
        fwd_ms = sum(op_ms) / max(n_fw, 1)
        # FLOP accounting.  `direct_tflops` prices a launch at the direct-convolution count 2*Cin*Cout*9*H*W.  The kernel
        # that runs is Winograd F(2,3) along x: 4 multiplications per output pair and kernel row instead of 6, i.e. 2/3 of those
        # FLOPs go through the matrix cores (the input/output transforms are VALU adds).  `achieved` is what the MFMA pipe executes --
        # the figure a roofline against the MFMA peak is about; the direct-equivalent rate is reported next to it.
        direct_tflops = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        wino = os.environ.get("PH_CONV_WINO", "1") != "0"
        mfma_share = 2.0 / 3.0 if wino else 1.0
        achieved = direct_tflops * mfma_share
        # HBM traffic of the conv launches: measured with rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
        # correction, WRITE_SIZE; tools/summarize_pmc.py) on this same command and committed under
        # profiles/; bench.py itself cannot read PMCs, so it reports the newest committed figure.
        traffic, traffic_src = None, None
        import glob

        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_conv_traffic.json")))
        if cands and B == 32:
            try:
                tj = json.load(open(cands[-1]))["conv3x3_mfma"]
                if int(round(tj["launches_per_forward"])) == len(conv_rows):
                    traffic, traffic_src = tj["hbm_bytes_per_launch"], os.path.relpath(cands[-1], ROOT)
            except Exception:
                pass
        conv_bytes = sum(r["bytes"] for r, _ in conv_rows)
        frames_total = B * world * args.steps
        res = {
            "metric": "frames/sec bottom-up UNet 1024x1024 inference",
            "value": frames_total / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg3: bottom-up UNet f16/r2/max_stride32/output_stride4, 1024x1024x1 uint8 frames, 13 nodes / 12 edges",
                "frames_per_gpu_per_step": B, "global_batch": B * world, "parallelism": f"dp{world} (frames sharded, no collective)",
                "weights": "xavier-uniform seed 1234, head x0.05", "postprocess_input": "rendered heads, 6 instances/frame (BASELINE.md s3)",
                "params": model.num_parameters(), "conv_gflop_per_frame": sum(r["flops"] for r in model.op_table(1, SIZE, SIZE)) / 1e9,
            },
            "roofline": {
                "bound": "mfma",
                "kernel": (f"conv3x3_wino_persist_kernel<64|32> (Winograd F(2,3) along x, " if wino else f"conv3x3_mfma_dma_persist_kernel<64|32> (direct, ")
                + f"{len(conv_rows)} launches/forward; the first encoder block runs in the fused stem kernel)",
                "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS,
                "flop_accounting": "achieved = FLOPs the MFMA pipe executes (Winograd: 2/3 of the direct-convolution count); "
                                   "direct_equivalent_* = direct-convolution FLOPs (2*Cin*Cout*9*H*W) / time",
                "direct_equivalent_tflops": direct_tflops, "direct_equivalent_frac": direct_tflops / MFMA_F32_PEAK_TFLOPS,
                "traffic": traffic, "traffic_unit": "HBM bytes per conv launch (PMC, avg over the launches of one forward)",
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": conv_bytes / max(len(conv_rows), 1),
                "algorithmic_gflop_per_forward": conv_flops * mfma_share / 1e9, "direct_gflop_per_forward": conv_flops / 1e9, "kernel_ms_per_forward": conv_ms,
                "avg_launch_ms": conv_ms / max(len(conv_rows), 1), "launches_per_forward": len(conv_rows),
                "conv_stack_direct_equivalent_tflops": stack_flops / (stack_ms * 1e-3) / 1e12 if stack_ms > 0 else 0.0,
                "conv_stack_direct_equivalent_frac": (stack_flops / (stack_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS) if stack_ms > 0 else 0.0,
                "forward_ms": fwd_ms, "forward_frames_per_s": B / (fwd_ms * 1e-3) if fwd_ms > 0 else 0.0,
                "per_op_ms": {r["label"]: round(ms / max(n_fw, 1), 4) for r, ms in zip(table, op_ms)},
                "profiled_forwards": n_fw,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(model.state_dict(), cms[:2].cpu(), pafs[:2].cpu())
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
