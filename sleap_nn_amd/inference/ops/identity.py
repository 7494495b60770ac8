"""Multi-class bottom-up grouping (sleap_nn/inference/ops/identity.py:13-146): class-map
sampling on the GPU (``ph_sample_class_maps``), Hungarian matching per (sample, node) on the
host (``ph_group_class_peaks``)."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
import torch

from sleap_nn_amd import _lib as L


def sample_class_maps(class_maps: torch.Tensor, peak_points: torch.Tensor, peak_sample_inds: torch.Tensor) -> torch.Tensor:
    """(n, n_classes) class probabilities at round-half-even(peak) clamped to the map."""
    L.require_cuda(class_maps, "class_maps")
    cm = class_maps.detach().to(torch.float32).contiguous()
    B, K, H, W = cm.shape
    n = int(peak_points.shape[0])
    out = torch.empty((n, K), dtype=torch.float32, device=cm.device)
    if n == 0:
        return out
    xy = peak_points.to(cm.device, torch.float32).contiguous()
    si = peak_sample_inds.to(cm.device, torch.int32).contiguous()
    with torch.cuda.device(cm.device):
        L.check(L.lib().ph_sample_class_maps(C.c_void_p(cm.data_ptr()), B, K, H, W, C.c_void_p(xy.data_ptr()), C.c_void_p(si.data_ptr()), n,
                                             C.c_void_p(out.data_ptr()), L.current_stream_ptr()))
    return out


def group_class_peaks(peak_class_probs, peak_sample_inds, peak_channel_inds, n_samples: int, n_channels: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """identity.py:13-76 -> (peak_inds, class_inds), int64 CPU tensors."""
    probs = np.ascontiguousarray(torch.as_tensor(peak_class_probs).detach().cpu().numpy(), dtype=np.float32)
    sb = np.ascontiguousarray(torch.as_tensor(peak_sample_inds).detach().cpu().numpy(), dtype=np.int32)
    sc = np.ascontiguousarray(torch.as_tensor(peak_channel_inds).detach().cpu().numpy(), dtype=np.int32)
    n = probs.shape[0]
    K = probs.shape[1] if probs.ndim == 2 else 0
    if n == 0 or K == 0:
        return torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64)
    pi = np.zeros(n, dtype=np.int32)
    ci = np.zeros(n, dtype=np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    cnt = L.lib().ph_group_class_peaks(p(probs), p(sb), p(sc), n, int(n_samples), int(n_channels), K, p(pi), p(ci))
    if cnt == L.PH_E_INFEASIBLE:
        raise ValueError("cost matrix is infeasible")
    L.check(cnt)
    return torch.from_numpy(pi[:cnt].astype(np.int64)), torch.from_numpy(ci[:cnt].astype(np.int64))


def classify_peaks_from_maps(class_maps, peak_points, peak_vals, peak_sample_inds, peak_channel_inds, n_channels: int):
    """identity.py:79-146 -> (points (B,K,N,2), point_vals (B,K,N), class_probs (B,K,N)), NaN = missing (CPU)."""
    B, K = int(class_maps.shape[0]), int(class_maps.shape[1])
    probs = sample_class_maps(class_maps, peak_points, peak_sample_inds).cpu()
    pts = peak_points.detach().cpu().to(torch.float32)
    vals = peak_vals.detach().cpu().to(torch.float32)
    sb = peak_sample_inds.detach().cpu().long()
    sc = peak_channel_inds.detach().cpu().long()
    pi, ci = group_class_peaks(probs, sb, sc, B, n_channels)
    points = torch.full((B, K, n_channels, 2), float("nan"))
    point_vals = torch.full((B, K, n_channels), float("nan"))
    class_probs = torch.full((B, K, n_channels), float("nan"))
    points[sb[pi], ci, sc[pi]] = pts[pi]
    point_vals[sb[pi], ci, sc[pi]] = vals[pi]
    class_probs[sb[pi], ci, sc[pi]] = probs[pi, ci]
    return points, point_vals, class_probs


def get_class_inds_from_vectors(peak_class_probs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """identity.py:149-173: Hungarian matching of samples (crops of ONE frame) to classes on
    ``-prob``; samples beyond the number of classes stay unassigned (-1 / NaN).  Host code (``ph_lsap``)."""
    from sleap_nn_amd.inference.ops.paf import linear_sum_assignment

    probs = torch.as_tensor(peak_class_probs).detach().cpu().to(torch.float32)
    n = int(probs.shape[0])
    inds = torch.full((n,), -1, dtype=torch.int64)
    pr = torch.full((n,), float("nan"))
    if n == 0 or probs.shape[1] == 0:
        return inds, pr
    r, c = linear_sum_assignment(-probs.numpy().astype(np.float64))
    for a, b in zip(r, c):
        inds[int(a)] = int(b)
        pr[int(a)] = probs[int(a), int(b)]
    return inds, pr
