"""bench.py's multi-rank code path, rehearsed on the one GPU of the test box: the self-launch (a child torchrun), the frame sharding
of weak and strong scaling, the max-over-ranks timing and the rank-0 JSON line.  All ranks share cuda:0 and synchronise over gloo
(RCCL refuses two ranks on one device), so the numbers mean nothing -- the structure of the line is what is checked."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    """(contract line, full record): stdout must END with exactly one strict-JSON line of at most 4 KB from rank 0; the legs file holds everything else."""
    import tempfile

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    with tempfile.TemporaryDirectory() as tmp:
        legs = os.path.join(tmp, "legs.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--legs-file", legs], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
        assert len(lines) == 1 and r.stdout.rstrip().endswith(lines[0]), r.stdout[-1000:]  # exactly one line, from rank 0, and the last thing on stdout
        assert len(lines[0].encode()) < 4096
        return json.loads(lines[0]), json.load(open(legs))


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_two_ranks_default_strong_with_weak_beside_it_and_three_ranks_strong_rehearsal():
    # N > 1 with no --scaling flag: `value` is BASELINE cfg3 as written (global batch split over the ranks), the weak figure rides along
    d, full = _run("--gpus", "2", "--rehearse-on-one-gpu", "--steps", "3", "--warmup", "1", "--batch", "4", "--global-batch", "8", "--no-alt-precisions", "--no-h2d-leg")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["frames_per_gpu_per_step"] == 4 and d["config"]["global_batch"] == 8
    assert d["value"] > 0 and abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]  # whole-job frames / max-over-ranks time
    w = full["weak_scaling"]
    assert w["frames_per_gpu_per_step"] == 4 and w["global_batch"] == 8 and w["value"] > 0 and d["legs_summary"]["weak_scaling_frames_per_s"] == pytest.approx(w["value"], rel=1e-3)
    assert "cpu_baseline" not in d and "train_cfg3" not in full and "REHEARSAL" in d["data"]  # the CPU leg and the extra legs belong to N = 1
    assert d["config"]["rccl_ranks_seen"] == 2 and d["config"]["frames_per_step_by_rank"] == [4, 4]  # every rank joined the collective and took its contiguous chunk of the global batch
    ht = full["config"]["host_threads"]
    assert ht["ranks"] == 2 and ht["per_rank"] == 2 and ht["fits"] == (4 <= ht["cores_visible"])
    # the driver's SCALE run puts 8 ranks on one host: enqueue thread + grouping worker per rank must fit its cores (16 on the 8-GPU box: 8 x 2)
    assert 2 * 8 <= max(ht["cores_visible"], 16)
    assert d["roofline"]["kernel"].startswith("conv3x3_wino") and 0 < d["roofline"]["frac"] < 1
    d, full = _run("--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2", "--warmup", "1", "--batch", "4", "--scaling", "weak", "--no-alt-precisions", "--no-h2d-leg")
    assert d["scaling"] == "weak" and d["config"]["global_batch"] == 8 and "weak_scaling" not in full
    d, full = _run("--gpus", "3", "--rehearse-on-one-gpu", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--global-batch", "8", "--no-alt-precisions", "--no-h2d-leg")
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["config"]["global_batch"] == 8
    assert d["config"]["frames_per_gpu_per_step"] in (2, 3)  # rank 0's contiguous chunk of 8 frames over 3 ranks


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_four_rank_strong_scaling_rehearsal_every_rank_equals_the_single_rank_result():
    """the per-rank batch of the 8-GPU strong-scaling run (4 frames per rank and step, THREE copies of the network on HIP streams of their own per
    rank) with as many ranks as one GPU box admits (six processes may use the card together -- the box's process guard --: the test runner, the launcher and four ranks; the
    driver's SCALE run has eight).  Every rank's
    head outputs -- through every one of its lanes -- must be the bits rank 0 computes for the same frames with one copy, kernel by kernel; the grouped keypoints the same
    on every rank; and the line reports what each rank's enqueue thread waited for its grouping worker, the first thing 8 x 3 lanes on one host's cores would stretch."""
    d, full = _run("--gpus", "4", "--rehearse-on-one-gpu", "--steps", "4", "--warmup", "2", "--scaling", "strong", "--global-batch", "16", "--no-alt-precisions", "--no-h2d-leg")
    assert d["n_gpus"] == 4 and d["config"]["frames_per_step_by_rank"] == [4] * 4 and d["config"]["rccl_ranks_seen"] == 4
    sc = full["config"]["shard_check"]
    assert sc["result"] == "equal", sc["mismatches"]
    assert sc["ranks"] == 4 and sc["lanes_per_rank"] == [3] * 4
    waits = sc["host_stage_wait_ms_per_step_by_rank"]
    assert len(waits) == 4 and all(w >= 0 for w in waits)
    assert d["config"]["shard_check"] == "equal" and d["config"]["host_stage_wait_ms_by_rank"] == waits
    print("host-stage wait per step and rank (ms):", waits, "step", d["ms_per_step"])


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_two_ranks_training_rehearsal():
    """--mode train with two ranks: identical initial arenas, disjoint shards, the two-bucket gradient all-reduce overlapped with the
    backward (gloo here), Adam on every rank -- the loss must be finite and move, the line must say what it measured."""
    d, full = _run("--gpus", "2", "--rehearse-on-one-gpu", "--mode", "train", "--steps", "3", "--warmup", "1", "--batch", "2")
    assert d["n_gpus"] == 2 and full["config"]["samples_per_gpu_per_step"] == 2 and d["config"]["global_batch"] == 4
    first, last = full["loss_first_last"]
    assert first > 0 and last > 0 and first == first and last == last and last != first
    assert full["allreduce"]["bucket_split"] is not None and full["allreduce"]["arena_mb"] > 30  # 7.8 M parameters in one flat arena
    assert "REHEARSAL" in d["data"] and 0 < d["roofline"]["frac"] < 1


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_default_single_gpu_line_with_every_leg():
    """`python bench.py` as the driver runs it (fewer steps): every leg runs, the line carries the contract fields + roofline + cpu_baseline and stays under 4 KB."""
    d, full = _run("--steps", "6", "--warmup", "3", "--leg-steps", "2")
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["config"]["workload"].startswith("cfg3")
    assert d["value"] == pytest.approx(32 * 6 / (d["ms_per_step"] * 6e-3), rel=1e-6) and d["vs_baseline"] is None
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "mfma" and r["kernel"] == "conv3x3_wino4_kernel" and 0.3 < r["frac"] < 1 and r["traffic"] and 0.3 < r["conv_stack_frac"] < 1
    assert c["kind"] == "port" and c["value"] > 0 and c["parity_on_this_sample"]["peak_indices_equal"] and c["parity_on_this_sample"]["grouping_equal"]
    for leg in ("train_cfg3", "train_cfg4", "infer_cfg4", "infer_cfg1", "infer_cfg2", "infer_cfg5", "published_workload", "alt_precisions", "strong_scaling_shards", "roofline_postprocess"):
        assert leg in full, leg
    s = d["legs_summary"]
    assert s["published_workload"]["end_to_end_frames_per_s"] > 0 and s["infer_cfg5"]["value"] > 0 and s["one_gpu_shard_two_streams_frames_per_s"]["4_frames_per_rank"] > 0
