"""Data-parallel sharding of a frame stream across the GPUs of one node.

The hot path has no cross-frame state, so multi-GPU inference is *replicas + sharding*: one
process per GPU (``torch.distributed``; backend ``nccl`` = RCCL on ROCm, ``gloo`` in CPU tests),
every rank holds the full weights (31 MB at cfg3), rank r processes the contiguous chunk
``[r*ceil(n/G), (r+1)*ceil(n/G))`` of each global batch, and results are gathered on the host in
rank order.  No collective sits on the data path; the gather moves only the final NaN-padded
keypoints.  The reference has no multi-GPU inference (predictor.py:925-931).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from sleap_nn_amd.inference.outputs import Outputs


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced (sizes differ by at most 1) chunk of ``range(n_items)`` for ``rank``."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _pad_instances(t: torch.Tensor, n_inst: int) -> torch.Tensor:
    if t.shape[1] == n_inst:
        return t
    pad = torch.full((t.shape[0], n_inst - t.shape[1], *t.shape[2:]), float("nan"), dtype=t.dtype)
    return torch.cat([t, pad], dim=1)


def gather_outputs(local: Optional[Outputs], group=None) -> Optional[List[Outputs]]:
    """Gather per-rank ``Outputs`` (slimmed, host tensors) to rank 0 in rank order."""
    payload = local.slim() if local is not None else None
    if not dist.is_available() or not dist.is_initialized():
        return [payload]
    world = dist.get_world_size(group)
    gathered = [None] * world if dist.get_rank(group) == 0 else None
    dist.gather_object(payload, gathered, dst=0, group=group)
    return gathered


def merge_outputs(parts: List[Optional[Outputs]]) -> Outputs:
    """Concatenate rank-ordered shards along the batch axis, NaN-padding the instance axis."""
    parts = [p for p in parts if p is not None and p.pred_keypoints is not None and p.pred_keypoints.shape[0] > 0]
    if not parts:
        return Outputs()
    n_inst = max(p.pred_keypoints.shape[1] for p in parts)
    kp = torch.cat([_pad_instances(p.pred_keypoints, n_inst) for p in parts])
    vals = torch.cat([_pad_instances(p.pred_peak_values, n_inst) for p in parts])
    scores = None
    if all(p.instance_scores is not None for p in parts):
        scores = torch.cat([_pad_instances(p.instance_scores, n_inst) for p in parts])
    return Outputs(pred_keypoints=kp, pred_peak_values=vals, instance_scores=scores)


def predict_sharded(layer, frames, group=None) -> Optional[Outputs]:
    """Run ``layer.predict`` on this rank's shard of ``frames`` (B, ...) and return the merged
    result on rank 0 (``None`` elsewhere)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    a, b = shard_bounds(len(frames), world, rank)
    local = layer.predict(frames[a:b]) if b > a else None
    parts = gather_outputs(local, group)
    return merge_outputs(parts) if rank == 0 else None


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """DDP gradient semantics on one contiguous arena: sum over ranks (one collective), divide by
    the world size.  With the ``nccl`` backend on ROCm this is a single RCCL all-reduce over xGMI."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
    return flat


class Communicator:
    """An RCCL communicator of the C ABI (``ph_comm_*``, include/posehip.h): created collectively by the ranks of a ``torch.distributed`` group from the 128-byte id rank 0
    draws -- ``torch.distributed`` only carries those bytes to the other ranks (its own nccl = RCCL backend or gloo; the gradient exchange itself is ``ph_allreduce`` /
    ``ph_model_set_comm``, no torch collective).  One per process and device; ``close()`` destroys it."""

    def __init__(self, handle, world: int, rank: int) -> None:
        self.handle, self.world, self.rank = handle, world, rank

    @staticmethod
    def available() -> bool:
        from sleap_nn_amd import _lib as L

        return bool(L.lib().ph_comm_available())

    @classmethod
    def create(cls, device, group=None) -> "Communicator":
        import ctypes as C

        from sleap_nn_amd import _lib as L

        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            L.check(L.lib().ph_comm_unique_id(buf))
        if world > 1:
            holder = [bytes(buf)]
            dist.broadcast_object_list(holder, src=0, group=group)
            C.memmove(buf, holder[0], 128)
        with torch.cuda.device(device):
            h = L.lib().ph_comm_create(buf, world, rank)
        if not h:
            raise RuntimeError("ph_comm_create failed: " + L.lib().ph_last_error().decode("utf-8", "replace"))
        return cls(h, world, rank)

    def all_reduce_(self, flat: torch.Tensor, stream=None) -> torch.Tensor:
        """In-place SUM of a contiguous fp32 device tensor over the ranks, on ``stream`` (default: the current one)."""
        import ctypes as C

        from sleap_nn_amd import _lib as L

        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else L.current_stream_ptr()
        with torch.cuda.device(flat.device):
            L.check(L.lib().ph_allreduce(self.handle, C.c_void_p(flat.data_ptr()), flat.numel(), sp))
        return flat

    def close(self) -> None:
        h, self.handle = self.handle, None
        if h:
            from sleap_nn_amd import _lib as L

            L.lib().ph_comm_destroy(h)


def all_reduce_buckets_(flat: torch.Tensor, split: Optional[int], group=None, tail_ready=None, comm_stream=None) -> float:
    """Sum ``flat`` over the ranks as two buckets -- the tail ``[split:]`` first, then the head ``[:split]`` -- and return the
    factor (1 / world) that turns the sum into DDP's mean (it is folded into the optimizer kernel, not applied here).

    On the GPU the caller passes the event the backward records when the tail is final (``tail_ready``) and a side stream:
    the tail's collective then overlaps the rest of the backward (``TrainingModule.all_reduce_grads``).  On host tensors
    (gloo, CPU tests) the two collectives simply run in that order; the result is the same as one all-reduce of the arena."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return 1.0
    world = dist.get_world_size(group)
    if split is None or split <= 0 or split >= flat.numel():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world
    if flat.is_cuda and comm_stream is not None:
        main = torch.cuda.current_stream(flat.device)
        with torch.cuda.stream(comm_stream):
            if tail_ready is not None:
                comm_stream.wait_event(tail_ready)
            else:
                comm_stream.wait_stream(main)
            dist.all_reduce(flat[split:], op=dist.ReduceOp.SUM, group=group)
            comm_stream.wait_stream(main)  # the rest of the backward
            dist.all_reduce(flat[:split], op=dist.ReduceOp.SUM, group=group)
        main.wait_stream(comm_stream)
    else:
        dist.all_reduce(flat[split:], op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(flat[:split], op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world
