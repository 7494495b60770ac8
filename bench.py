"""Benchmark of the hot path: bottom-up UNet 1024x1024 inference (BASELINE.json cfg3).

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--mode infer|train]

One *step* (``--mode infer``, the headline) = one pass of the hot path over one per-GPU batch of synthetic
frames: uint8 frames -> UNet forward (random-init weights of the cfg3 architecture, every kernel a hand-written
gfx950 kernel, replayed as one hipGraph) -> local peaks + integral refinement -> PAF line scoring -> D2H of the
scored candidates -> C++ matching/assembly -> NaN-padded keypoints on the host.  Because random weights give
meaningless peaks, the post-process stage consumes *rendered* heads with 6 animals/frame
that are equally resident in HBM; forward and post-process both run in full inside every step, back to back on
one stream.

``value`` is measured with the frames already resident in HBM (the bench contract).  The same number of steps is
then timed a second time with the uint8 frames starting in pinned HOST memory, copied asynchronously on a copy
stream under the previous step's kernels (double-buffered) -- SURVEY section 8(d)'s H2D-inclusive frame time; it
is reported as ``h2d_inclusive`` and is never ``value``.

Multi-GPU: one process per GPU (torch.distributed / RCCL).  ``python bench.py --gpus N`` without a launcher
starts the N ranks ITSELF (a child ``python -m torch.distributed.run``, before this process touches the GPU)
and fails if the job ends up with a different number of ranks.  Frames are sharded across ranks, no data-path
collective (frames are independent).  At N > 1 the default is ``--scaling strong`` = BASELINE cfg3 as written ("batch=32,
DP across 8xMI355X"): ``--global-batch`` (32) frames per step split into contiguous chunks of 32/G per
rank; the weak-scaling figure (``--batch`` = 32 frames per GPU per step) is measured in the same launch and reported beside
it as ``weak_scaling``.  ``--scaling weak`` makes the weak figure ``value`` (at N = 1 the two coincide).
value = total frames / max-over-ranks time.

Prints ONE JSON line on rank 0 with the contract fields plus ``roofline`` (dominant kernel = the MFMA conv3x3,
HIP-event timed inside the timed region), ``step_ms`` (median / p10 / p90 of the per-step GPU time) and, at N=1,
``cpu_baseline`` (the oracle on the host cores at 1 and at all threads, with the parity of the HIP path against it
on the very frames it timed).

The default N = 1 run also carries three short extra legs, each with its own ``roofline`` (executed / direct FLOP
accounting): ``train_cfg3`` (the cfg3 network's training step, 32 frames), ``train_cfg4`` (BASELINE cfg4: ConvNeXt-tiny
centered-instance, 64 crops of 384x384, forward + MSE + backward + Adam) and ``infer_cfg4`` (the same network's inference
forward, 64 crops).  They run after the headline's timed regions and never touch ``value``.

``--mode train`` times data-parallel training steps (forward + MSE + backward + gradient all-reduce + Adam) of the
cfg3 UNet or the cfg4 ConvNeXt-tiny (``--train-config``) as the headline of the line; see run_train().
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from benchlegs.common import (CFG3_BB, CFG3_HEADS, CFG4_BB, CFG4_HEADS, MFMA_F16_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS, NODES, SIZE, _cfg5_traffic, _forward_profile, _matrix_rows,  # noqa: F401  (re-exported: tests and tools import them from bench)
                              _pad16, _small_roofline, _time_calls, conv_kernel_long_names, conv_kernel_short_names, forward_executed_flops, percentiles, rendered_heads, synthetic_instances)
from benchlegs.infer_cfg4 import infer_cfg4_leg
from benchlegs.infer_cfg5 import infer_cfg5_leg
from benchlegs.published import PUBLISHED_A40, PUBLISHED_LANES, published_workload_leg  # noqa: F401
from benchlegs.single_instance import SI_BB, single_instance_leg  # noqa: F401
from benchlegs.train import train_leg


def cpu_baseline(model, layer, cms_dev, pafs_dev, dev, budget_s: float = 24.0):
    """Oracle (torch-CPU restatement of the reference, kind = "port") on the host cores, on the same weights, the same
    frame and the same rendered heads as the GPU leg -- and the parity of the HIP path against it on exactly those
    inputs (SURVEY section 8d "CPU baseline timing").  The only place bench.py touches oracle/."""
    from oracle import cpu_ref as O
    from sleap_nn_amd.inference.ops.peaks import find_local_peaks
    from sleap_nn_amd.inference.preprocess_info import PreprocInfo

    sd = model.state_dict()
    g = torch.Generator().manual_seed(4321)
    img = torch.randint(0, 256, (1, 1, SIZE, SIZE), dtype=torch.uint8, generator=g)
    cms, pafs = cms_dev[:2].cpu(), pafs_dev[:2].cpu()
    scorer = O.PAFScorerRef(NODES, [tuple(e) for e in CFG3_HEADS["pafs"]["edges"]], 8)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def timed(threads, fwd_budget, post_budget, max_n):
        torch.set_num_threads(threads)
        with torch.inference_mode():
            O.model_forward(sd, CFG3_BB, CFG3_HEADS, "bottomup", img[:, :, :256, :256])  # thread-pool warm-up
            t_f, n_f, ref = 0.0, 0, None
            t0 = time.perf_counter()
            while n_f < 2 or (time.perf_counter() - t0 < fwd_budget and n_f < max_n):
                t = time.perf_counter()
                ref = O.model_forward(sd, CFG3_BB, CFG3_HEADS, "bottomup", img)
                t_f += time.perf_counter() - t
                n_f += 1
            t_p, n_p, post = 0.0, 0, None
            t0 = time.perf_counter()
            while n_p < 2 or (time.perf_counter() - t0 < post_budget and n_p < max_n):
                t = time.perf_counter()
                post = O.bottomup_postprocess(cms, pafs, scorer, 4)
                t_p += time.perf_counter() - t
                n_p += 1
        return t_f / n_f, t_p / (n_p * cms.shape[0]), n_f, n_p, ref, post

    # all threads: pick the best of a few pool sizes with one short probe each (oneDNN degrades when oversubscribed on big hosts)
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
    f_n, p_n, n_f, n_p, ref, post = timed(best_t, budget_s * 0.45, budget_s * 0.1, 20)
    f_1, p_1, n_f1, n_p1, _, _ = timed(1, budget_s * 0.35, budget_s * 0.1, 4)
    torch.set_num_threads(best_t)

    # ---- parity of the HIP path on the same inputs
    out = model.forward(img.to(dev))
    d_cms = float((out["MultiInstanceConfmapsHead"].cpu() - ref["MultiInstanceConfmapsHead"]).abs().max())
    d_paf = float((out["PartAffinityFieldsHead"].cpu() - ref["PartAffinityFieldsHead"]).abs().max())
    rp, rv, rb, rc = O.find_local_peaks(cms, 0.2, None)
    gp, gv, gb, gc = [t.cpu() for t in find_local_peaks(cms_dev[:2], 0.2, None)]
    peaks_equal = bool(gp.shape == rp.shape and torch.equal(gp, rp) and torch.equal(gv, rv) and torch.equal(gb, rb) and torch.equal(gc, rc))
    got = layer.postprocess({"MultiInstanceConfmapsHead": cms_dev[:2], "PartAffinityFieldsHead": pafs_dev[:2]}, PreprocInfo(eff_scale=torch.ones(2)))
    rk, rvals, rs = post
    k = got.pred_keypoints.numpy()
    grouping_equal = bool(k.shape == rk.shape and np.array_equal(np.isnan(k), np.isnan(rk)) and np.allclose(k, rk, atol=1e-3, equal_nan=True)
                          and np.allclose(got.instance_scores.numpy(), rs, atol=1e-4, equal_nan=True))
    return {
        "value": 1.0 / (f_n + p_n), "unit": "frames/s", "cores": best_t, "kind": "port",
        "value_1thread": 1.0 / (f_1 + p_1),
        "sample": f"{n_f} forwards of 1 frame 1024x1024 ({1e3 * f_n:.0f} ms each at {best_t} threads; {n_f1} at 1 thread: {1e3 * f_1:.0f} ms) + "
                  f"{n_p} post-process passes over 2 rendered frames ({1e3 * p_n:.1f} ms/frame); oracle/cpu_ref.py, torch-CPU fp32",
        "parity_on_this_sample": {"max_abs_confmap_diff": d_cms, "max_abs_paf_diff": d_paf,
                                  "confmap_abs_max": float(ref["MultiInstanceConfmapsHead"].abs().max()), "paf_abs_max": float(ref["PartAffinityFieldsHead"].abs().max()), "peak_indices_equal": peaks_equal,
                                  "n_peaks": int(rp.shape[0]), "grouping_equal": grouping_equal, "n_instances": int((~np.isnan(rs)).sum())},
    }


PROFILE_MIN_SAMPLES = 6  # forwards of the timed region that launch kernel by kernel with per-op HIP events


def profile_steps(steps: int) -> list:
    """The steps of a timed region whose forward is event-profiled: never step 0 or 1 (right behind the barrier, on an idle GPU), every
    max(2, steps // PROFILE_MIN_SAMPLES)-th step from step 2 on -- six samples at the driver's --steps 20 as at the default 50."""
    return list(range(2, steps, max(2, steps // PROFILE_MIN_SAMPLES)))


def summarize_op_samples(samples, stack_idx, step_ms_median):
    """Per-op MEDIAN over the profiled forwards (a sample = the per-op milliseconds of ONE forward), and the check that makes the figure usable: the summed conv-stack
    time of the medians may exceed the median step time by at most 3 % (kernel-by-kernel launches carry a boundary per op the hipGraph replay does not); beyond that
    the sampling was disturbed (a cold first sample, a host stall between launches) and the result says so instead of passing for a roofline."""
    if not samples:
        return {"op_ms": [], "n": 0, "sampling": "none", "stack_ms": 0.0, "stack_ms_by_sample": []}
    a = np.asarray(samples, dtype=np.float64)
    med = np.median(a, axis=0)
    stack = float(med[list(stack_idx)].sum()) if len(stack_idx) else 0.0
    ok = step_ms_median is None or step_ms_median <= 0 or stack <= 1.03 * step_ms_median
    return {"op_ms": [float(v) for v in med], "n": int(a.shape[0]), "sampling": "ok" if ok else "inflated", "stack_ms": stack,
            "stack_ms_by_sample": [round(float(r[list(stack_idx)].sum()), 4) for r in a]}


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> int:
    """Start the N ranks as a CHILD torchrun job (this process has not touched the GPU; it only waits)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


CONTRACT_LINE_MAX = 4096
LEG_KEYS = ("train_cfg3", "train_cfg4", "infer_cfg4", "infer_cfg1", "infer_cfg2", "infer_cfg5", "published_workload", "alt_precisions",
            "strong_scaling_shards", "weak_scaling", "roofline_postprocess")


def _num(x, nd=4):
    """Round a float to `nd` significant-ish decimals for the contract line (None / non-finite stay None: strict JSON has no NaN)."""
    if x is None:
        return None
    if isinstance(x, bool) or isinstance(x, int):
        return x
    x = float(x)
    if not np.isfinite(x):
        return None
    return float(f"{x:.{nd + 2}g}")


def _pick(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def legs_summary(full: dict) -> dict:
    """One number (or a short tuple of numbers) per extra leg: what the line says about them; the legs themselves live in the legs file."""
    s = {}
    for k in ("train_cfg3", "train_cfg4", "infer_cfg4", "infer_cfg1", "infer_cfg2", "infer_cfg5"):
        if isinstance(full.get(k), dict):
            leg = full[k]
            s[k] = {"value": _num(leg.get("value")), "unit": leg.get("unit"), "ms_per_step": _num(leg.get("ms_per_step")), "mfma_frac": _num(_pick(leg, "roofline", "frac"))}
            if _pick(leg, "two_streams", "value") is not None:
                s[k]["two_streams"] = _num(_pick(leg, "two_streams", "value"))
            if k == "infer_cfg5":  # HBM bytes per forward (PMC evidence file named in the legs file)
                s[k]["traffic"] = _num(_pick(leg, "roofline", "traffic"))
    pw = full.get("published_workload")
    if isinstance(pw, dict):
        s["published_workload"] = {"forward_ms_per_batch4": _num(pw.get("value")), "forward_frames_per_s": _num(pw.get("frames_per_s_forward")),
                                   "end_to_end_frames_per_s": _num(_pick(pw, "end_to_end", "value")), "topdown_end_to_end_frames_per_s": _num(_pick(pw, "topdown", "end_to_end_fps")),
                                   "single_instance_end_to_end_frames_per_s": _num(_pick(pw, "single_instance", "end_to_end_fps"))}
    alt = full.get("alt_precisions")
    if isinstance(alt, dict):
        s["alt_precisions_frames_per_s"] = {k: _num(v.get("value")) for k, v in alt.items() if isinstance(v, dict)}
    sh = full.get("strong_scaling_shards")
    if isinstance(sh, dict):
        s["one_gpu_shard_frames_per_s"] = {k: _num(v.get("value")) for k, v in sh.items() if isinstance(v, dict)}
        if any(isinstance(v, dict) and "two_streams" in v for v in sh.values()):
            s["one_gpu_shard_two_streams_frames_per_s"] = {k: _num(_pick(v, "two_streams", "value")) for k, v in sh.items() if isinstance(v, dict)}
    ws = full.get("weak_scaling")
    if isinstance(ws, dict):
        s["weak_scaling_frames_per_s"] = _num(ws.get("value"))
    pp = full.get("roofline_postprocess")
    if isinstance(pp, dict):
        s["peaks_kernel"] = {"bound": "hbm", "achieved_GBps": _num(pp.get("achieved")), "peak_GBps": pp.get("peak"), "frac": _num(pp.get("frac")),
                             "us_per_batch": _num(pp.get("us_per_batch")), "traffic": _num(pp.get("traffic"))}
    return s


def contract_line(full: dict, legs_file: str | None = None) -> str:
    """The ONE stdout line of the bench contract, built from the full record: only the contract fields + `roofline` + `cpu_baseline` +
    a one-number-per-leg summary.  Strict JSON (no NaN / Infinity), at most CONTRACT_LINE_MAX bytes; raises if it cannot be met, so a
    record that would not parse on the driver's side never leaves this process looking like a success."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: (_num(full[k], 6) if isinstance(full.get(k), float) else full.get(k)) for k in keep if k in full}
    cfg = full.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "frames_per_gpu_per_step", "global_batch", "parallelism", "inputs", "forward_launch", "rccl_ranks_seen",
                                          "frames_per_step_by_rank", "params", "crops_per_gpu_per_step", "samples_per_gpu_per_step") if k in cfg}
    if isinstance(cfg.get("conv_gflop_per_frame"), float):
        line["config"]["conv_gflop_per_frame"] = _num(cfg["conv_gflop_per_frame"])
    if isinstance(cfg.get("shard_check"), dict):  # N > 1: every rank's outputs against one rank computing the same frames; the ranks' host-stage waits
        line["config"]["shard_check"] = cfg["shard_check"].get("result")
        line["config"]["host_stage_wait_ms_by_rank"] = cfg["shard_check"].get("host_stage_wait_ms_per_step_by_rank")
    if isinstance(full.get("step_ms"), dict):
        line["step_ms"] = {k: _num(v) for k, v in full["step_ms"].items()}
    if isinstance(full.get("h2d_inclusive"), dict):
        line["h2d_inclusive"] = {"value": _num(full["h2d_inclusive"].get("value"), 6), "unit": full["h2d_inclusive"].get("unit"), "ms_per_step": _num(full["h2d_inclusive"].get("ms_per_step")),
                                 "what": "SURVEY 8(d) frame time: the same steps with the uint8 frames starting in pinned host memory"}
    r = full.get("roofline")
    if isinstance(r, dict):
        rl = {"bound": r.get("bound"), "kernel": str(r.get("kernel", "")).split(" (")[0].split(",")[0], "achieved": _num(r.get("achieved")), "peak": r.get("peak"), "unit": r.get("unit"),
              "frac": _num(r.get("frac")), "traffic": _num(r.get("traffic"), 6)}
        for k_out, k_in in (("traffic_source", "traffic_source"), ("algorithmic_bytes_per_launch", "algorithmic_bytes_per_launch"), ("avg_launch_ms", "dominant_kernel_avg_launch_ms"),
                            ("launches_per_forward", "launches_per_forward"), ("dominant_kernel_ms_per_forward", "dominant_kernel_ms_per_forward"),
                            ("executed_gflop_per_forward", "algorithmic_gflop_per_forward"), ("direct_gflop_per_forward", "direct_gflop_per_forward"),
                            ("conv_kernel_ms_per_forward", "kernel_ms_per_forward"), ("forward_ms", "forward_ms"), ("direct_equivalent_tflops", "direct_equivalent_tflops"),
                            ("conv_stack_frac", "conv_stack_frac"), ("conv_stack_ms_per_forward", "conv_stack_ms_per_forward"), ("executed_gflop_per_step", "executed_gflop_per_step")):
            if r.get(k_in) is not None:
                rl[k_out] = _num(r[k_in], 6) if isinstance(r[k_in], float) else r[k_in]
        if isinstance(r.get("kernels"), dict):
            rl["kernels"] = {k.split("<")[0]: {"launches": e.get("launches_per_forward"), "ms": _num(e.get("ms_per_forward")), "frac": _num(e.get("frac_of_peak"))} for k, e in r["kernels"].items()}
        if r.get("sampling") is not None:
            rl["sampling"] = r["sampling"]  # "ok" | "inflated": median of >= 6 event-profiled forwards against the median step (section 5)
        rl["accounting"] = "achieved = FLOPs the MFMA pipe EXECUTES in `kernel`'s launches (Winograd: 1/4 or 4/9 of the direct count) / their HIP-event time in the timed region"
        line["roofline"] = rl
    c = full.get("cpu_baseline")
    if isinstance(c, dict):
        line["cpu_baseline"] = {"value": _num(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"), "value_1thread": _num(c.get("value_1thread")),
                                "sample": str(c.get("sample", ""))[:260], "parity_on_this_sample": {k: (_num(v) if isinstance(v, float) else v) for k, v in (c.get("parity_on_this_sample") or {}).items()}}
    summ = legs_summary(full)
    if summ:
        line["legs_summary"] = summ
    if legs_file:
        line["legs_file"] = legs_file
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text.encode()) > CONTRACT_LINE_MAX:  # prose goes first, then the per-leg summary: never the contract fields
        line.get("roofline", {}).pop("accounting", None)
        line.get("h2d_inclusive", {}).pop("what", None)
        if "cpu_baseline" in line:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:120]
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text.encode()) > CONTRACT_LINE_MAX and "legs_summary" in line:
        line["legs_summary"] = {"dropped": "summary did not fit; see legs_file"}
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text.encode()) > CONTRACT_LINE_MAX:
        raise RuntimeError(f"bench.py: contract line is {len(text.encode())} bytes > {CONTRACT_LINE_MAX}")
    json.loads(text)
    return text


def _json_safe(o):
    """The full record with non-finite floats as null, so the legs file is strict JSON too."""
    if isinstance(o, dict):
        return {str(k): _json_safe(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_json_safe(v) for v in o]
    if isinstance(o, (float, np.floating)):
        return float(o) if np.isfinite(o) else None
    if isinstance(o, np.integer):
        return int(o)
    return o


def emit(res: dict, legs_file: str | None = None) -> None:
    """Full record -> legs file (+ one line on stderr); contract line -> stdout, LAST, flushed, nothing after it."""
    if legs_file is None:
        d = os.path.join(ROOT, "gpurun_out")
        legs_file = os.path.join(d if os.path.isdir(d) else ROOT, "bench_legs.json")
    full = _json_safe(res)
    rel = None
    try:
        with open(legs_file, "w") as f:
            json.dump(full, f, allow_nan=False, indent=1)
        rel = os.path.relpath(legs_file, ROOT)
    except OSError as e:  # a read-only tree must not cost the bench line
        print(f"bench.py: could not write {legs_file}: {e}", file=sys.stderr)
    print("bench.py full record: " + json.dumps(full, allow_nan=False), file=sys.stderr)
    sys.stderr.flush()
    sys.stdout.flush()
    print(contract_line(full, rel), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=32, help="frames per step over all GPUs (strong scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong at N > 1 (BASELINE cfg3: global batch 32 split over the GPUs; the weak figure rides along as "
                         "`weak_scaling`), weak at N = 1 (the same thing there)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the train_cfg3 / train_cfg4 / infer_cfg4 legs of the default N = 1 run")
    ap.add_argument("--leg-steps", type=int, default=6, help="timed steps of each extra leg")
    ap.add_argument("--small-legs-only", action="store_true", help="print only the small-batch legs (infer_cfg1 / infer_cfg2 / published_workload) as one JSON object: a development shortcut, not the bench line")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer")
    ap.add_argument("--train-config", choices=["cfg3", "cfg4"], default="cfg3")
    ap.add_argument("--dtype", choices=["f32", "f16x3", "f16"], default="f32",
                    help="arithmetic of the 3x3 convolutions: f32 = exact fp32 MFMA (the headline); f16x3 = split-fp16 (22-bit products, fp32 accumulate); "
                         "f16 = the reference's autocast mode.  The other two are timed as short extra legs of the default run (alt_precisions)")
    ap.add_argument("--no-alt-precisions", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the forward kernel by kernel instead of replaying one hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--legs-file", default=None,
                    help="where the FULL record (every leg, every per-op table) goes; default gpurun_out/bench_legs.json when gpurun_out/ exists, "
                         "else bench_legs.json next to bench.py.  stdout carries only the <= 4 KB contract line")
    ap.add_argument("--no-h2d-leg", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="REHEARSAL of the multi-rank code path on a one-GPU box: all ranks share cuda:0 and synchronise over gloo "
                         "(RCCL refuses two ranks on one device).  The line it prints is marked as such and is not a measurement.")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    args.scaling_defaulted = args.scaling is None
    if args.scaling is None:
        if args.mode == "train":  # cfg4 IS a global batch of 64 (BASELINE); the cfg3 training leg keeps the reference's DP semantics (per-GPU batch fixed)
            args.scaling = "strong" if (args.train_config == "cfg4" and args.gpus > 1) else "weak"
        else:
            args.scaling = "strong" if args.gpus > 1 else "weak"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if torch.cuda.device_count() < world and not (args.rehearse_on_one_gpu and torch.cuda.device_count() >= 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPUs are visible")
    import torch.distributed as dist

    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and args.rehearse_on_one_gpu:
        dist.init_process_group(backend="gloo")
    elif world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: RCCL sees {dist.get_world_size()} ranks, --gpus asked for {args.gpus}")
        # ranks share the host: keep torch's CPU pool (synthetic-input rendering only) to a fair share
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    ctx = {"rank": rank, "world": world, "dev": dev, "dist": dist}
    if args.small_legs_only:
        print(json.dumps(small_batch_legs(args, ctx)))
        return
    res = run_train(args, ctx) if args.mode == "train" else run_infer(args, ctx)
    if rank == 0:
        if args.rehearse_on_one_gpu:
            res["data"] = "REHEARSAL: %d ranks on one GPU over gloo -- not a measurement" % world
        emit(res, args.legs_file)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _peaks_traffic(conv_traffic_json, batch):
    """HBM bytes per find_local_peaks call (both launches) from the PMC passes of tools/run_profile.sh (FETCH_SIZE x 2 + WRITE_SIZE, per launch): bench.py cannot read
    PMCs and quotes the newest committed file, which is for the 32-frame step."""
    if not conv_traffic_json or batch != 32:
        return None
    try:
        ks = json.load(open(conv_traffic_json))["kernels"]
        tot = 0.0
        for k, e in ks.items():
            if "peaks_onepass" in k or "peaks_place" in k:
                tot += e.get("fetch_bytes_per_launch", 0.0) + e.get("write_bytes_per_launch", 0.0)
        return tot or None
    except Exception:
        return None


def warm_up(fn, min_calls: int, min_ms: float = None, finish=None) -> None:
    """Untimed calls in front of a timed loop: at least `min_calls`, and for at least `min_ms` (default benchlegs.common.WARM_MS) of wall time, so that the loop does not start at the
    clocks of a GPU that idled through the host-side set-up before it."""
    from benchlegs.common import WARM_MS

    lim = WARM_MS if min_ms is None else min_ms
    t_w, k = time.perf_counter(), 0
    while k < min_calls or 1e3 * (time.perf_counter() - t_w) < lim:
        fn()
        k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()
    if finish is not None:
        finish()
    torch.cuda.synchronize()


def run_infer(args, ctx):
    rank, world, dev, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["dist"]
    from concurrent.futures import ThreadPoolExecutor

    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.inference.backends import HipBackend
    from sleap_nn_amd.inference.layers import BottomUpLayer
    from sleap_nn_amd.inference.ops.paf import PAFScorer
    from sleap_nn_amd.inference.preprocess_info import PreprocInfo
    from sleap_nn_amd.inference.streaming import group_scored_batch
    from sleap_nn_amd.parallel import shard_bounds

    def shard(scaling):
        """(frames of this rank per step, global batch, pinned host frames) for one scaling mode."""
        if scaling == "strong":
            lo, hi = shard_bounds(args.global_batch, world, rank)  # rank-contiguous chunk of the global batch
            if hi - lo <= 0:
                raise SystemExit(f"--global-batch {args.global_batch} leaves rank {rank} of {world} without frames")
            g = torch.Generator().manual_seed(4321)
            allf = torch.randint(0, 256, (max(args.global_batch, 1), 1, 1, SIZE, SIZE), dtype=torch.uint8, generator=g)
            return hi - lo, args.global_batch, allf[lo:hi].contiguous().pin_memory()
        g = torch.Generator().manual_seed(4321 + rank)
        return args.batch, args.batch * world, torch.randint(0, 256, (args.batch, 1, 1, SIZE, SIZE), dtype=torch.uint8, generator=g).pin_memory()

    B, global_batch, host_frames = shard(args.scaling)
    PREC = {"f32": "exact", "f16x3": "split", "f16": "fp16"}
    precision = PREC[args.dtype]
    fp16 = precision != "exact"  # the conv stack runs on the fp16 matrix pipe
    model = Model("unet", CFG3_BB, CFG3_HEADS, "bottomup")
    model.init_xavier_(seed=1234, head_scale=0.05)  # the reference's xavier_init_weights; heads x0.05 keep outputs O(1)
    use_graph = not args.no_graph
    backend = HipBackend(model, str(dev), use_graph=use_graph, precision=precision)
    eager = HipBackend(model, str(dev), precision=precision) if use_graph else backend  # same model handle: the profiled steps launch kernel by kernel
    layer = BottomUpLayer(backend, PAFScorer.from_config(CFG3_HEADS), 4, 8, max_stride=32)
    frames = host_frames.to(dev)
    if use_graph:  # the resident frames live in the graph's own input buffer (HipBackend.static_input): a step's forward is one graph launch, no staging copy in front
        frames = backend.static_input((B, 1, SIZE, SIZE)).copy_(frames.squeeze(1))
    cms, pafs = rendered_heads(B, dev)
    info = PreprocInfo(original_size=(SIZE, SIZE), processed_size=(SIZE, SIZE), eff_scale=torch.ones(B), output_stride=4)

    pool = ThreadPoolExecutor(max_workers=1)
    params = layer.grouping_params()
    pending, inflight = [], []
    # Multi-GPU strong scaling leaves a rank a few frames per step (4 at 8 GPUs): most launches of such a step have fewer work units than CUs.  Consecutive steps then alternate
    # between TWO (8 frames) or THREE (4 frames) copies of the network on HIP streams of their own (as Predictor.from_model_paths does for small networks): +3.5 % at 8 frames, +8 % at 4 on one GPU
    # (strong_scaling_shards.*.two_streams).  Never at N = 1 / 32 frames per step: there every launch fills the chip.
    lanes = [(backend, layer, None)]
    if world > 1 and use_graph and B <= 8:
        from sleap_nn_amd.inference.predictor import concurrent_streams

        n_lanes = SHARD_LANES_4 if B <= 4 else 2
        lane_st = concurrent_streams(dev, n_lanes)  # (two streams on ONE hardware queue would run in order: chosen by a measured overlap)
        lanes = [(backend, layer, lane_st[0])]
        for k in range(1, n_lanes):
            model_b = Model("unet", CFG3_BB, CFG3_HEADS, "bottomup")
            model_b.init_xavier_(seed=1234, head_scale=0.05)
            backend_b = HipBackend(model_b, str(dev), use_graph=True, precision=precision)
            lanes.append((backend_b, BottomUpLayer(backend_b, PAFScorer.from_config(CFG3_HEADS), 4, 8, max_stride=32), lane_st[k]))
    step_no, last_stream, host_wait = [0], [None], [0.0]
    split_marks = None  # (diagnostic: PH_BENCH_SPLIT=1 records an event behind the forward of the first timed steps)

    heads_in = {"cms": cms, "pafs": pafs, "info": info}  # rebound for the weak-scaling leg of a multi-GPU run

    def step(x, profiled=False):
        """Forward + peaks + PAF scoring + async D2H are enqueued for this batch; then the PREVIOUS batch's
        results (its D2H event fired long ago) are handed to the C++ grouping worker.  The GPU always has the
        next batch queued and the grouping of batch k-1 overlaps the GPU work of batch k
        (Predictor._predict_streaming_pipelined).  Every batch is grouped before the closing barrier."""
        be_k, layer_k, stream_k = lanes[0 if (profiled or len(lanes) == 1 or x.shape[0] > 8) else step_no[0] % len(lanes)]
        step_no[0] += 1
        last_stream[0] = stream_k if stream_k is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(stream_k) if stream_k is not None else contextlib.nullcontext():
            raw = (eager if profiled else be_k)(x)  # uint8 frames -> {"MultiInstanceConfmapsHead", "PartAffinityFieldsHead"}
            if split_marks is not None and len(split_marks) < 8:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                split_marks.append(ev)
            inflight.append((layer_k, layer_k._enqueue_scoring({"MultiInstanceConfmapsHead": heads_in["cms"], "PartAffinityFieldsHead": heads_in["pafs"]}, heads_in["info"])))
        if len(inflight) > len(lanes):
            ly, h = inflight.pop(0)
            pending.append(pool.submit(group_scored_batch, ly._finish_scoring(h), params))
        out = None
        if len(pending) > 1:
            tw = time.perf_counter()
            out = pending.pop(0).result()
            host_wait[0] += time.perf_counter() - tw  # the enqueue thread blocked on this rank's grouping worker
        return raw, out

    def drain():
        while inflight:
            ly, h = inflight.pop(0)
            pending.append(pool.submit(group_scored_batch, ly._finish_scoring(h), params))
        outs = [f.result() for f in pending]
        pending.clear()
        return outs

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, 2)):
        raw, out = step(frames, profiled=(i == 0))
    out = drain()[-1]
    torch.cuda.synchronize()
    n_inst = int((~torch.isnan(out.instance_scores)).sum())
    assert n_inst >= 5 * B, f"post-process found only {n_inst} instances in {B} frames"
    assert all(torch.isfinite(v).all() for v in raw.values())

    # ---- timed region 1 (the contract's `value`): frames resident in HBM.  HIP-event timing of every op runs INSIDE it, on
    # every PROFILE_EVERY-th step (those steps launch the forward kernel by kernel with ~26 event records, the others replay
    # the hipGraph); one event per step on the compute stream gives the per-step GPU times.
    model.set_profiling(True)
    model.set_profiling(False)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    prof_at = set(profile_steps(args.steps))
    op_samples, sample_pending = [], False

    def collect_sample():  # the events of the previous profiled forward completed steps ago: reading them stalls nothing
        ms, n = model.read_profile()
        if n > 0:
            op_samples.append([v / n for v in ms])

    # The contract's W warm-up steps ran above; everything the host had to do since (asserts with a device read-back, event creation) left the GPU idle for milliseconds, and the
    # first two steps of the timed region then ran 1.5 - 1.7 ms slow each (`step_ms_profiled_vs_replayed.by_step` of a 20-step run: 12.39, 12.14, 10.83, 10.67 ...: the clocks of an
    # idle GPU).  Two more UNTIMED steps close that gap: the synchronisation below ends microseconds before the first timed launch, as the contract's bracket wants it.
    for _ in range(2):
        step(frames)
    drain()  # (host-side grouping of those steps: a millisecond or two with nothing on the GPU ...)
    for _ in range(2):
        backend(frames)  # (... so two bare forwards run right up to the synchronisation: nothing of the step pipeline is left in flight, and the gap to the first timed launch is microseconds)
    torch.cuda.synchronize()
    barrier()
    host_wait[0] = 0.0
    if os.environ.get("PH_BENCH_SPLIT"):
        split_marks = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        prof = i in prof_at
        if prof:
            if sample_pending:
                collect_sample()
            model.set_profiling(True)  # (clears: one sample = one forward)
            sample_pending = True
        else:
            model.set_profiling(False, resume=True)
        step(frames, profiled=prof)
        marks[i + 1].record()
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    if sample_pending:
        collect_sample()
    model.set_profiling(False)
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if split_marks:
        print("forward / rest of the first timed steps (ms):", [(round(marks[i].elapsed_time(split_marks[i]), 3), round(split_marks[i].elapsed_time(marks[i + 1]), 3)) for i in range(min(len(split_marks), args.steps))], file=sys.stderr)
        split_marks = None
    host_wait_ms = 1e3 * host_wait[0] / max(args.steps, 1)

    # ---- N > 1: what every rank computed, checked against ONE rank computing the same frames (rank 0 redoes each rank's chunk with its first lane): the head outputs of the
    # rank's frames through EVERY lane (its copies of the network on their own streams) and the grouped keypoints of its last step, bit for bit; plus the time each rank's
    # enqueue thread spent blocked on its grouping worker -- the first place 8 ranks x 3 lanes x (1 enqueue thread + 1 C++ worker) collide on a host
    shard_check = None
    if world > 1:
        import hashlib

        def digest(t):
            return hashlib.sha1(t.detach().to("cpu").contiguous().numpy().tobytes()).hexdigest()

        def heads_digest(be, x):
            r = be(x)
            torch.cuda.synchronize()
            return [digest(r[k]) for k in sorted(r)]

        xin = host_frames.to(dev).squeeze(1)
        mine = {"rank": rank, "frames": B, "lanes": [heads_digest(be_k, xin) for be_k, _ly, _st in lanes], "keypoints": digest(torch.nan_to_num(out.pred_keypoints)) if out is not None else None,
                "host_wait_ms": host_wait_ms}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        if rank == 0:
            ok, why = True, []
            for g_ in gathered:
                if args.scaling == "strong":
                    lo_, hi_ = shard_bounds(args.global_batch, world, g_["rank"])
                    gg = torch.Generator().manual_seed(4321)
                    fr = torch.randint(0, 256, (max(args.global_batch, 1), 1, 1, SIZE, SIZE), dtype=torch.uint8, generator=gg)[lo_:hi_]
                else:
                    gg = torch.Generator().manual_seed(4321 + g_["rank"])
                    fr = torch.randint(0, 256, (args.batch, 1, 1, SIZE, SIZE), dtype=torch.uint8, generator=gg)
                want = heads_digest(eager, fr.to(dev).squeeze(1))  # kernel by kernel on rank 0's first copy
                for li_, got in enumerate(g_["lanes"]):
                    if got != want:
                        ok = False
                        why.append(f"rank {g_['rank']} lane {li_}: head outputs differ from the single-rank forward of its frames")
                if g_["frames"] == gathered[0]["frames"] and g_["keypoints"] != gathered[0]["keypoints"]:
                    ok = False
                    why.append(f"rank {g_['rank']}: grouped keypoints differ from rank 0's")
            shard_check = {"result": "equal" if ok else "DIFFERENT", "what": "sha1 of each rank's head outputs (every lane) vs rank 0's kernel-by-kernel forward of the same frames; grouped keypoints of the last step vs rank 0's",
                           "ranks": world, "lanes_per_rank": [len(g_["lanes"]) for g_ in gathered], "mismatches": why[:8],
                           "host_stage_wait_ms_per_step_by_rank": [round(g_["host_wait_ms"], 4) for g_ in gathered]}
    _tab0 = model.op_table(B, SIZE, SIZE)
    from sleap_nn_amd import _lib as _L0

    op_summary = summarize_op_samples(op_samples, [j for j, r in enumerate(_tab0) if r["kind"] in (_L0.OP_CONV, _L0.OP_STEM, _L0.OP_INPUT_CONV)],
                                      float(np.median([m for j, m in enumerate(step_ms) if j not in prof_at])) if len(step_ms) > len(prof_at) else None)
    op_ms, n_fw = (op_summary["op_ms"] or [0.0] * len(_tab0)), 1

    # ---- timed region 2: the same steps with the uint8 frames coming from pinned host memory (H2D on a copy stream, double
    # buffered, overlapped with the previous step's kernels)
    elapsed_h2d = None
    if not args.no_h2d_leg:
        copy_stream = torch.cuda.Stream(dev)
        bufs = [torch.empty_like(frames), torch.empty_like(frames)]
        landed = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]

        def upload(k):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(consumed[k & 1])  # the step that last read this buffer is done with it
                bufs[k & 1].copy_(host_frames.view(bufs[0].shape), non_blocking=True)
                landed[k & 1].record(copy_stream)

        for e in consumed:
            e.record()
        upload(0)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            upload(i + 1)  # rides under step i's kernels
            for _be, _ly, st in lanes:  # (whichever stream runs the step)
                (st if st is not None else torch.cuda.current_stream()).wait_event(landed[i & 1])
            step(bufs[i & 1])
            consumed[i & 1].record(last_stream[0])
        drain()
        barrier()
        elapsed_h2d = time.perf_counter() - t0

    # ---- N > 1, default scaling: `value` above is the strong-scaling figure (BASELINE cfg3: global batch 32 split over the
    # GPUs); the weak-scaling figure (--batch frames per GPU per step) is measured here, in the same launch
    elapsed_weak, weak_B = None, None
    if world > 1 and args.scaling == "strong" and args.scaling_defaulted:
        weak_B, weak_global, weak_host = shard("weak")
        wframes = weak_host.to(dev)
        wcms, wpafs = rendered_heads(weak_B, dev)
        heads_in.update(cms=wcms, pafs=wpafs, info=PreprocInfo(original_size=(SIZE, SIZE), processed_size=(SIZE, SIZE), eff_scale=torch.ones(weak_B), output_stride=4))
        for _ in range(max(args.warmup, 2)):
            step(wframes)
        drain()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(wframes)
        drain()
        barrier()
        elapsed_weak = time.perf_counter() - t0
        heads_in.update(cms=cms, pafs=pafs, info=info)
        del wframes, wcms, wpafs

    # ---- extra legs (N = 1, default run only): the same steps with the convolution stack on the fp16 matrix pipe
    alt = {}
    if world == 1 and precision == "exact" and not args.no_alt_precisions:
        ref_heads = {k: v.clone() for k, v in eager(frames[:2]).items()}
        for tag, prec in (("f16x3_split", "split"), ("f16_autocast", "fp16")):
            model.set_precision(prec)
            warm_up(lambda: step(frames), 3, finish=drain)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step(frames)
            drain()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            got = eager(frames[:2])
            alt[tag] = {"value": B * args.steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / args.steps,
                        "max_abs_head_diff_vs_exact": {k: float((got[k] - ref_heads[k]).abs().max()) for k in ref_heads},
                        "head_abs_max": {k: float(ref_heads[k].abs().max()) for k in ref_heads}}
        model.set_precision("exact")
        alt["note"] = ("same steps, 3x3 convolutions on v_mfma_f32_32x32x16_f16: f16x3_split = operands as (hi, lo) fp16 pairs, 3 MFMAs per product, fp32 "
                       "accumulation (22-bit products; parity tests hold it to the same 1e-4 bar as the exact path); f16_autocast = the reference's autocast mode "
                       "(tolerance 5e-3).  Not part of `value`.")

    # ---- the local-maxima kernels on their own (north_star's "wavefront-reduction local-maxima kernel ... HBM GB/s"): HIP events around the two launches, on the step's rendered maps
    peaks_us = None
    if rank == 0:
        from sleap_nn_amd.inference.ops.peaks import find_local_peaks_device

        pc = layer.postprocess_config
        # (five warm-up calls, not a time-based warm-up: this leg runs right behind the steps above, as the kernel does in the pipeline; 20 ms of NOTHING BUT these 26-us launches lets the chip
        # drop its memory-side clocks -- the same 50 calls then take 40 us each instead of 26)
        for _ in range(5):
            find_local_peaks_device(cms, pc.peak_threshold, pc.effective_refinement, pc.integral_patch_size, 4096, xy_scale=4.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            find_local_peaks_device(cms, pc.peak_threshold, pc.effective_refinement, pc.integral_patch_size, 4096, xy_scale=4.0)
        e1.record()
        torch.cuda.synchronize()
        peaks_us = 1e3 * e0.elapsed_time(e1) / 50

    # ---- N = 1 default run: the per-rank batches of the strong-scaling runs (global batch 32 over 8 / 4 GPUs = 4 / 8 frames per rank), measured on this one GPU with the
    # same pipelined step -- what one rank of the driver's SCALE run does per step, without the other ranks
    shards = {}
    if world == 1 and precision == "exact" and not args.no_extra_legs and args.scaling_defaulted and B >= 8:
        for sb in (8, 4):
            scms, spafs = rendered_heads(sb, dev)
            heads_in.update(cms=scms, pafs=spafs, info=PreprocInfo(original_size=(SIZE, SIZE), processed_size=(SIZE, SIZE), eff_scale=torch.ones(sb), output_stride=4))
            sframes = frames.reshape(B, 1, SIZE, SIZE)[:sb].contiguous()
            if use_graph:
                sframes = backend.static_input((sb, 1, SIZE, SIZE)).copy_(sframes)
            warm_up(lambda: step(sframes), 5, finish=drain)
            n_s = max(args.steps, 50)
            t1 = time.perf_counter()
            for _ in range(n_s):
                step(sframes)
            drain()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            shards[f"{sb}_frames_per_rank"] = {"value": sb * n_s / dt, "unit": "frames/s per GPU", "ms_per_step": 1e3 * dt / n_s, "steps": n_s, "stands_for": f"one rank of --gpus {32 // sb} --scaling strong"}
            # the same steps alternating between TWO copies of the network on two HIP streams (what Predictor.from_model_paths does for small networks, with three): a step of a few
            # frames leaves CUs idle in most launches, two independent steps in flight fill them
            if use_graph:
                if "lane2" not in heads_in:
                    from sleap_nn_amd.inference.predictor import concurrent_streams

                    extra = []
                    for _ in range(max(SHARD_LANES_4, 2) - 1):
                        model2 = Model("unet", CFG3_BB, CFG3_HEADS, "bottomup")
                        model2.init_xavier_(seed=1234, head_scale=0.05)
                        backend2 = HipBackend(model2, str(dev), use_graph=True, precision=precision)
                        extra.append((backend2, BottomUpLayer(backend2, PAFScorer.from_config(CFG3_HEADS), 4, 8, max_stride=32)))
                    heads_in["lane2"] = (extra, concurrent_streams(dev, max(SHARD_LANES_4, 2)))
                extra, lanes_st = heads_in["lane2"]
                n_copies = SHARD_LANES_4 if sb <= 4 else 2  # what a rank of bench.py --gpus N runs with (lanes above)
                shard_lanes = [(backend, layer, sframes)] + [(be2, ly2, be2.static_input((sb, 1, SIZE, SIZE)).copy_(sframes)) for be2, ly2 in extra[: n_copies - 1]]
                fl, futs = [], []

                def step2(i):
                    k = i % n_copies
                    be_k, ly_k, x_k = shard_lanes[k]
                    with torch.cuda.stream(lanes_st[k]):
                        be_k(x_k)
                        fl.append((ly_k, ly_k._enqueue_scoring({"MultiInstanceConfmapsHead": heads_in["cms"], "PartAffinityFieldsHead": heads_in["pafs"]}, heads_in["info"])))
                    if len(fl) > n_copies:
                        ly0, h0 = fl.pop(0)
                        futs.append(pool.submit(group_scored_batch, ly0._finish_scoring(h0), params))
                    if len(futs) > 2:
                        futs.pop(0).result()

                def drain2():
                    while fl:
                        ly0, h0 = fl.pop(0)
                        futs.append(pool.submit(group_scored_batch, ly0._finish_scoring(h0), params))
                    for f in futs:
                        f.result()
                    futs.clear()

                torch.cuda.synchronize()
                for i in range(6):
                    step2(i)
                drain2()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(n_s):
                    step2(i)
                drain2()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                shards[f"{sb}_frames_per_rank"]["two_streams"] = {"value": sb * n_s / dt2, "unit": "frames/s per GPU", "ms_per_step": 1e3 * dt2 / n_s,
                                                                 "copies": n_copies,
                                                                 "what": "consecutive steps alternate between copies of the network on HIP streams of their own (`copies`)"}
        heads_in.update(cms=cms, pafs=pafs, info=info)
        heads_in.pop("lane2", None)
        del scms, spafs, sframes

    t = torch.tensor([elapsed, elapsed_h2d or 0.0, elapsed_weak or 0.0], dtype=torch.float64, device=dev)
    # multi-rank bookkeeping for the line: how many ranks the collective library actually joined (a sum of ones over the job) and what each rank processed per step
    ranks_seen, per_rank_frames = 1, [B]
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
        mine = torch.zeros(world, dtype=torch.float64, device=dev)
        mine[rank] = B
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank_frames = [int(v) for v in mine.tolist()]
    elapsed, elapsed_h2d = float(t[0].item()), (float(t[1].item()) if elapsed_h2d is not None else None)
    elapsed_weak = float(t[2].item()) if elapsed_weak is not None else None
    if rank != 0:
        return None

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
    # FLOP accounting.  `direct_tflops` prices a launch at the direct-convolution count 2*Cin*Cout*9*H*W.  The kernels
    # that run are Winograd forms: conv3x3_wino4_kernel (F(4x4,3x3): 36 multiplications per sixteen outputs instead of 144, 1/4 of those FLOPs go
    # through the matrix cores) for the N-tile-64 layers from 64 input channels on where its time model prefers it, conv3x3_wino2d_kernel<64> (F(2x2,3x3): 4/9) for the other
    # N-tile-64 layers (and the conv whose 1x1 head rides in its epilogue), conv3x3_w16_kernel (wave-private F(2x2,3x3): 4/9) for the Cout <= 32 ones; the input / output transforms are VALU work.  `achieved` is
    # what the MFMA pipe EXECUTES in the launches of the dominant kernel over their duration -- the figure a roofline against the
    # MFMA peak is about; the direct-equivalent rate is reported next to it and is NOT a roofline fraction.
    # Which kernel ran each conv launch is read back from the library (ph_model_last_kernels of an eager forward), not re-derived here.
    eager(frames)
    kv = model.last_kernels()
    torch.cuda.synchronize()
    KSHORT = conv_kernel_short_names()
    by_kernel = {}
    prev_key = None
    for (r, ms), code in zip(conv_rows, [c for row, c in zip(table, kv) if row["kind"] == L.OP_CONV]):
        if code == L.KV_FUSED and prev_key is not None:  # a conv computed inside the launch of the conv in front of it (block2_c32_f16_kernel): its work belongs to that launch
            e = by_kernel[prev_key]
            e["ms"] += ms
            e["direct_flops"] += r["flops"]
            e["executed_flops"] += r["flops"]
            e["bytes"] += r["bytes"]
            continue
        share = (3.0 if precision == "split" else 1.0) if code == L.KV_F16 else L.KV_MFMA_SHARE[code]
        prev_key = KSHORT[code]
        e = by_kernel.setdefault(KSHORT[code], {"launches": 0, "ms": 0.0, "direct_flops": 0.0, "executed_flops": 0.0, "bytes": 0.0})
        e["launches"] += 1
        e["ms"] += ms
        e["direct_flops"] += r["flops"]
        e["executed_flops"] += r["flops"] * share
        e["bytes"] += r["bytes"]
    dom = max(by_kernel, key=lambda k: by_kernel[k]["ms"]) if by_kernel else "direct"
    D = by_kernel.get(dom, {"launches": 0, "ms": 0.0, "direct_flops": 0.0, "executed_flops": 0.0, "bytes": 0.0})
    direct_tflops = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    achieved = D["executed_flops"] / (D["ms"] * 1e-3) / 1e12 if D["ms"] > 0 else 0.0
    executed_all = sum(e["executed_flops"] for e in by_kernel.values())
    peak = MFMA_F16_PEAK_TFLOPS if fp16 else MFMA_F32_PEAK_TFLOPS
    KERNEL_NAMES = conv_kernel_long_names(precision)
    # HBM traffic of the conv launches: measured with rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE;
    # tools/summarize_pmc.py) on this same command and committed under profiles/; bench.py itself cannot read PMCs, so it
    # reports the newest committed figure whose launch count matches this run.
    traffic, traffic_src = None, None
    import glob

    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_conv_traffic.json")))
    if B == 32 and not fp16:
        for cand in reversed(cands):  # newest first; files of another leg (the fp16 pipe's cfg5 set) or of another launch count are stepped over, not an error
            try:
                tj = json.load(open(cand))["conv3x3_mfma"]
                if int(round(tj["launches_per_forward"])) == len(conv_rows):
                    traffic, traffic_src = tj["hbm_bytes_per_launch"], os.path.relpath(cand, ROOT)
                    break
            except Exception:
                continue
    conv_bytes = sum(r["bytes"] for r, _ in conv_rows)
    frames_total = global_batch * args.steps
    res = {
        "metric": "frames/sec bottom-up UNet 1024x1024 inference",
        "value": frames_total / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": {"f32": "f32", "f16x3": "f16x3 (split-fp16 operand pairs, f32 accumulate)", "f16": "f16 (f32 accumulate)"}[args.dtype],
        "data": "synthetic",
        "config": {
            "workload": "cfg3: bottom-up UNet f16/r2/max_stride32/output_stride4, 1024x1024x1 uint8 frames, 13 nodes / 12 edges",
            "frames_per_gpu_per_step": B, "global_batch": global_batch, "parallelism": f"dp{world} (frames sharded, no collective)",
            "weights": "xavier-uniform seed 1234, head x0.05", "postprocess_input": "rendered heads, 6 instances/frame (BASELINE.md s3)",
            "params": model.num_parameters(), "conv_gflop_per_frame": sum(r["flops"] for r in model.op_table(1, SIZE, SIZE)) / 1e9,
            "forward_launch": "hipGraph replay (steps with per-op events launch kernel by kernel)" if use_graph else "kernel by kernel",
            "inputs": "uint8 frames resident in HBM when the timed region starts",
            "rccl_ranks_seen": ranks_seen, "frames_per_step_by_rank": per_rank_frames, "streams_per_rank": len(lanes), "shard_check": shard_check,
            "host_threads": {"cores_visible": (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)), "per_rank": 2, "ranks": world,
                             "what": "one Python thread that enqueues the GPU work (and the pinned H2D staging of the h2d_inclusive leg) + one C++ grouping worker per rank",
                             "fits": 2 * world <= (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))},
        },
        "step_ms": percentiles(step_ms),
        "untimed_steps_before_timing": max(args.warmup, 2) + 2,  # --warmup steps + two more (and two bare forwards) right in front of the timed region's synchronisation (the GPU would idle through the host-side checks otherwise and start the region at idle clocks)
        "step_ms_profiled_vs_replayed": {"profiled_mean": float(np.mean([v for i, v in enumerate(step_ms) if i in prof_at])) if prof_at else None,
                                         "replayed_mean": float(np.mean([v for i, v in enumerate(step_ms) if i not in prof_at])) if len(prof_at) < len(step_ms) else None,
                                         "by_step": [round(float(v), 3) for v in step_ms]},
        "roofline": {
            "bound": "mfma",
            "kernel": KERNEL_NAMES[dom] + f", {D['launches']} of the {len(conv_rows)} conv launches of a forward; the first encoder block runs in the fused stem kernel",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "flop_accounting": "achieved = FLOPs the MFMA pipe executes in the launches of `kernel` / their summed duration (HIP events inside the timed region); "
                               "direct_equivalent_tflops = direct-convolution FLOPs (2*Cin*Cout*9*H*W) of ALL conv launches / their time, a throughput figure, not a roofline fraction",
            "direct_equivalent_tflops": direct_tflops,
            "kernels": {KERNEL_NAMES[k].split(" ")[0]: {"launches_per_forward": e["launches"], "ms_per_forward": e["ms"], "direct_gflop": e["direct_flops"] / 1e9,
                                                          "executed_gflop": e["executed_flops"] / 1e9, "executed_tflops": e["executed_flops"] / (e["ms"] * 1e-3) / 1e12 if e["ms"] > 0 else 0.0,
                                                          "frac_of_peak": e["executed_flops"] / (e["ms"] * 1e-3) / 1e12 / peak if e["ms"] > 0 else 0.0,
                                                          "algorithmic_bytes": e["bytes"]} for k, e in by_kernel.items()},
            "traffic": traffic, "traffic_unit": "HBM bytes per conv launch (PMC, avg over the launches of one forward)",
            "traffic_source": traffic_src, "algorithmic_bytes_per_launch": conv_bytes / max(len(conv_rows), 1),
            "algorithmic_gflop_per_forward": executed_all / 1e9, "direct_gflop_per_forward": conv_flops / 1e9, "kernel_ms_per_forward": conv_ms,
            "avg_launch_ms": conv_ms / max(len(conv_rows), 1), "launches_per_forward": len(conv_rows),
            "dominant_kernel_ms_per_forward": D["ms"], "dominant_kernel_avg_launch_ms": D["ms"] / max(D["launches"], 1),
            "conv_stack_direct_equivalent_tflops": stack_flops / (stack_ms * 1e-3) / 1e12 if stack_ms > 0 else 0.0,
            # every conv launch of the forward incl. the fused stem (its MFMA conv priced by its own share; its VALU first conv is not matrix work)
            "conv_stack_frac": (executed_all + sum(r.get("mfma_flops", 0.0) * L.KV_MFMA_SHARE[L.KV_STEM] for r, _ in stack_rows if r["kind"] == L.OP_STEM)) / (stack_ms * 1e-3) / 1e12 / peak if stack_ms > 0 else 0.0,
            "conv_stack_ms_per_forward": stack_ms,
            "forward_ms": fwd_ms, "forward_frames_per_s": B / (fwd_ms * 1e-3) if fwd_ms > 0 else 0.0,
            "per_op_ms": {r["label"]: round(ms / max(n_fw, 1), 4) for r, ms in zip(table, op_ms)},
            "profiled_forwards": op_summary["n"], "sampling": op_summary["sampling"], "conv_stack_ms_by_sample": op_summary["stack_ms_by_sample"],
            "sampling_rule": "per-op MEDIAN over the profiled forwards (steps " + ",".join(str(v) for v in sorted(prof_at)) + " of the timed region; never steps 0 / 1); "
                             "'inflated' = the conv stack of the medians exceeds 1.03 x the median un-profiled step, the figure is then not a roofline",
        },
    }
    if peaks_us is not None:
        cm_bytes = float(cms.numel() * 4)
        peaks_traffic, peaks_traffic_src = None, None
        for cand in reversed(cands):  # the newest committed PMC file that holds the peak kernels
            peaks_traffic = _peaks_traffic(cand, B)
            if peaks_traffic:
                peaks_traffic_src = os.path.relpath(cand, ROOT)
                break
        res["roofline_postprocess"] = {"bound": "hbm", "kernel": "peaks_onepass_kernel<1> + peaks_place_kernel (find_local_peaks: threshold over the streamed maps, 3x3 strict NMS + integral refinement of the candidates, ordered placement)",
                                       "achieved": cm_bytes / (peaks_us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": cm_bytes / (peaks_us * 1e-6) / 1e9 / 8000.0,
                                       "algorithmic_bytes": cm_bytes, "us_per_batch": peaks_us, "launches": 2, "traffic": peaks_traffic, "traffic_source": peaks_traffic_src,
                                       "byte_accounting": "algorithmic bytes = the confidence maps read once (B x 13 x 256 x 256 fp32); time = HIP events around 50 back-to-back calls (both launches + the output allocation of the wrapper); "
                                                          "a bare read of the same 109 MB (tools/probes/hbm_read_probe.hip) takes 15.9 us on this GPU = 0.86 of 8 TB/s, launch included"}
    if elapsed_weak is not None:
        res["weak_scaling"] = {"value": weak_B * world * args.steps / elapsed_weak, "unit": "frames/s", "ms_per_step": 1e3 * elapsed_weak / args.steps,
                               "frames_per_gpu_per_step": weak_B, "global_batch": weak_B * world,
                               "note": "same launch, --batch frames per GPU per step (per-GPU work fixed as N grows); `value` above is the strong-scaling figure"}
    if alt:
        res["alt_precisions"] = alt
    if shards:
        shards["note"] = "one-GPU measurements of the per-rank batches the strong-scaling runs use (no other ranks, no barrier): not a scaling curve"
        res["strong_scaling_shards"] = shards
    if elapsed_h2d is not None:
        res["h2d_inclusive"] = {"value": frames_total / elapsed_h2d, "unit": "frames/s", "ms_per_step": 1e3 * elapsed_h2d / args.steps,
                                "note": "same steps, uint8 frames start in pinned host memory; async H2D on a copy stream, double-buffered under the previous step"}
    if world == 1 and not args.no_cpu_baseline and not fp16:
        res["cpu_baseline"] = cpu_baseline(model, layer, cms, pafs, dev)
    if world == 1 and not fp16 and not args.no_extra_legs and args.scaling_defaulted:
        del backend, eager, layer, model, frames, cms, pafs
        heads_in.clear()
        res.update(extra_legs(args, ctx))
        res.update(small_batch_legs(args, ctx))
    return res


SHARD_LANES_4 = 3     # copies of the cfg3 network a rank with <= 4 frames per step alternates between (8 frames: 2); whole pipelined step on one GPU, tools/shard_lanes_ab.py: 2 637 -> 2 671 frames/s


def small_batch_legs(args, ctx):
    """infer_cfg1 / infer_cfg2 / published_workload of the default N = 1 line (the small-batch regime, driver-timed)."""
    dev = ctx["dev"]
    out = {}
    torch.cuda.empty_cache()
    out["infer_cfg1"] = single_instance_leg("cfg1", 256, 1, 5, 400, dev, with_cpu=not args.no_cpu_baseline)
    out["infer_cfg2"] = single_instance_leg("cfg2", 512, 8, 13, 200, dev, with_cpu=False)
    torch.cuda.empty_cache()
    out["published_workload"] = published_workload_leg(200, dev)
    torch.cuda.empty_cache()
    out["infer_cfg5"] = infer_cfg5_leg(50, dev)
    torch.cuda.empty_cache()
    return out


def extra_legs(args, ctx):
    """train_cfg3 / train_cfg4 / infer_cfg4 of the default N = 1 line (driver-timed training and ConvNeXt numbers)."""
    dev = ctx["dev"]
    out = {}
    torch.cuda.empty_cache()
    res, model = train_leg("cfg3", 32, 32, args.leg_steps, 2, ctx)
    out["train_cfg3"] = res
    del model
    torch.cuda.empty_cache()
    res, model = train_leg("cfg4", 64, 64, args.leg_steps, 2, ctx)
    out["train_cfg4"] = res
    torch.cuda.empty_cache()
    out["infer_cfg4"] = infer_cfg4_leg(model, 64, max(args.leg_steps, 8), 2, dev)
    del model
    torch.cuda.empty_cache()
    for leg in out.values():  # the legs share the headline's launch: drop what only a stand-alone line needs
        for k in ("n_gpus", "warmup", "higher_is_better", "scaling", "vs_baseline"):
            leg.pop(k, None)
    return out


def run_train(args, ctx):
    """``--mode train``: the training step as the headline.  ``--train-config cfg3``: ``--batch`` frames per GPU (weak) or
    ``--global-batch`` split (strong); ``cfg4``: global batch 64 split over the ranks (strong, by definition of BASELINE cfg4;
    ``--scaling weak`` gives every rank 64 crops)."""
    rank, world = ctx["rank"], ctx["world"]
    from sleap_nn_amd.parallel import shard_bounds

    if args.train_config == "cfg4":
        gb = 64
        B = gb if args.scaling == "weak" else shard_bounds(gb, world, rank)[1] - shard_bounds(gb, world, rank)[0]
        global_batch = gb * world if args.scaling == "weak" else gb
    elif args.scaling == "strong":
        lo, hi = shard_bounds(args.global_batch, world, rank)
        B, global_batch = hi - lo, args.global_batch
    else:
        B, global_batch = args.batch, args.batch * world
    if B <= 0:
        raise SystemExit(f"rank {rank} of {world} has no samples")
    res, _ = train_leg(args.train_config, B, global_batch, args.steps, args.warmup, ctx, args.scaling)
    return res


if __name__ == "__main__":
    main()
