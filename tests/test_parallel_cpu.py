"""World-size-2 gloo test of the frame sharding + rank-ordered gather (no GPU needed: the layer is
a host-side stand-in; the data path itself has no collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.parallel import allreduce_mean_, merge_outputs, predict_sharded, shard_bounds


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeLayer:
    """Deterministic stand-in: frame value v -> v instances-worth of keypoints."""

    def predict(self, frames):
        B = frames.shape[0]
        n_inst = int(frames.reshape(B, -1)[:, 0].max().item()) + 1
        kp = torch.full((B, n_inst, 3, 2), float("nan"))
        for b in range(B):
            k = int(frames[b].reshape(-1)[0])
            kp[b, : k + 1] = float(k)
        return Outputs(pred_keypoints=kp, pred_peak_values=kp[..., 0].clone(), instance_scores=kp[..., 0, 0].clone())


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = torch.arange(7).reshape(7, 1, 1, 1).float()
    out = predict_sharded(_FakeLayer(), frames)
    g = torch.full((1000,), float(rank + 1))
    allreduce_mean_(g)  # gradient-arena semantics: mean over ranks
    assert torch.allclose(g, torch.full((1000,), 1.5))
    if rank == 0:
        q.put(out.pred_keypoints.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_predict_sharded_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    kp = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = _FakeLayer().predict(torch.arange(7).reshape(7, 1, 1, 1).float()).pred_keypoints.numpy()
    assert kp.shape == ref.shape
    assert np.array_equal(np.nan_to_num(kp, nan=-1), np.nan_to_num(ref, nan=-1))  # rank order == frame order


def test_merge_outputs_pads_instances():
    a = Outputs(pred_keypoints=torch.zeros(2, 1, 3, 2), pred_peak_values=torch.zeros(2, 1, 3), instance_scores=torch.zeros(2, 1))
    b = Outputs(pred_keypoints=torch.ones(1, 3, 3, 2), pred_peak_values=torch.ones(1, 3, 3), instance_scores=torch.ones(1, 3))
    m = merge_outputs([a, None, b])
    assert m.pred_keypoints.shape == (3, 3, 3, 2)
    assert torch.isnan(m.pred_keypoints[:2, 1:]).all() and (m.pred_keypoints[2] == 1).all()


def test_bench_launches_its_own_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with no launcher must start two rank processes itself (a child torchrun, before any GPU
    call) and must not quietly measure fewer GPUs than asked: on this GPU-less host the ranks report the shortfall and
    the command exits non-zero.  A mismatching WORLD_SIZE from an outer launcher is refused as well."""
    import os
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("needs a host without GPUs (on a GPU box this would run the benchmark)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    # every rank that gets to speak reports the shortfall (the launcher may tear the second one down as soon as the first has failed)
    assert 1 <= r.stderr.count("--gpus 2 but only 0 GPUs are visible") <= 2, r.stderr[-1500:]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def _dp_worker(rank, world, port, q):
    """One rank of a data-parallel training step on HOST tensors (gloo): the oracle supplies forward/backward, the product's
    bucketed all-reduce + the 1/world scale give DDP's averaged gradient, torch's Adam applies it."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import cpu_ref as O
    from sleap_nn_amd.parallel import all_reduce_buckets_, shard_bounds

    bb = {"in_channels": 1, "kernel_size": 3, "filters": 4, "filters_rate": 2, "max_stride": 4, "stem_stride": None, "middle_block": True, "up_interpolate": True,
          "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    heads = {"confmaps": {"part_names": ["a", "b"], "output_stride": 2, "loss_weight": 1.0}}
    sd = O.init_state(bb, heads, "single_instance", seed=5, head_scale=1.0)
    g = torch.Generator().manual_seed(9)
    img = torch.randint(0, 256, (4, 1, 16, 24), dtype=torch.uint8, generator=g)
    tgt = {"SingleInstanceConfmapsHead": torch.rand(4, 2, 8, 12, generator=g)}
    lo, hi = shard_bounds(4, world, rank)  # disjoint, equal shards (DistributedSampler semantics)
    _, grads = O.training_step(sd, bb, heads, "single_instance", img[lo:hi], {k: v[lo:hi] for k, v in tgt.items()}, [1.0])
    keys = list(sd.keys())
    flat = torch.cat([grads[k].reshape(-1) for k in keys])
    split = sum(sd[k].numel() for k in keys[: len(keys) // 2])  # any clean boundary: the result must not depend on it
    scale = all_reduce_buckets_(flat, split)
    whole = torch.cat([grads[k].reshape(-1) for k in keys])
    dist.all_reduce(whole)
    q.put((rank, (flat * scale).numpy(), (whole / world).numpy(), scale))
    dist.destroy_process_group()


def test_data_parallel_gradient_semantics_world2_gloo():
    """Two ranks with disjoint halves of a batch: bucketed all-reduce x (1 / world) == DDP's mean gradient == the gradient of the
    global batch on one rank (MSE is a mean over equal shards), independent of where the bucket boundary sits."""
    from oracle import cpu_ref as O

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][3] == 0.5
    assert np.array_equal(res[0][1], res[1][1]) and np.allclose(res[0][1], res[0][2], rtol=0, atol=0)
    bb = {"in_channels": 1, "kernel_size": 3, "filters": 4, "filters_rate": 2, "max_stride": 4, "stem_stride": None, "middle_block": True, "up_interpolate": True,
          "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    heads = {"confmaps": {"part_names": ["a", "b"], "output_stride": 2, "loss_weight": 1.0}}
    sd = O.init_state(bb, heads, "single_instance", seed=5, head_scale=1.0)
    g = torch.Generator().manual_seed(9)
    img = torch.randint(0, 256, (4, 1, 16, 24), dtype=torch.uint8, generator=g)
    tgt = {"SingleInstanceConfmapsHead": torch.rand(4, 2, 8, 12, generator=g)}
    _, full = O.training_step(sd, bb, heads, "single_instance", img, tgt, [1.0])
    ref = torch.cat([full[k].reshape(-1) for k in sd.keys()]).numpy()
    assert np.allclose(res[0][1], ref, rtol=1e-5, atol=1e-8)
