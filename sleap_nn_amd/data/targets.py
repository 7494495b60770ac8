"""Training targets rendered on the GPU.

``generate_multiconfmaps`` / ``generate_pafs`` keep the reference's call shapes
(``sleap_nn/data/confidence_maps.py:46-94``, ``sleap_nn/data/edge_maps.py:250-323``) but take a whole
batch ``(B, I, N, 2)`` and run one kernel instead of a Python loop over instances per sample.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import torch

from sleap_nn_amd import _lib as L


def _grid(size: int, stride: int) -> int:
    return (size + stride - 1) // stride


def generate_multiconfmaps(instances: torch.Tensor, img_hw: Tuple[int, int], sigma: float = 1.5, output_stride: int = 2) -> torch.Tensor:
    """``instances``: (B, I, N, 2) (centroids: (B, I, 2)) on the GPU, NaN = missing -> (B, N, h, w)."""
    L.require_cuda(instances, "instances")
    pts = instances.unsqueeze(-2) if instances.dim() == 3 else instances
    pts = pts.to(torch.float32).contiguous()
    B, I, N, _ = pts.shape
    h, w = _grid(img_hw[0], output_stride), _grid(img_hw[1], output_stride)
    out = torch.empty((B, N, h, w), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        L.check(L.lib().ph_render_confmaps(C.c_void_p(pts.data_ptr()), B, I, N, int(img_hw[0]), int(img_hw[1]), int(output_stride), float(sigma),
                                           C.c_void_p(out.data_ptr()), L.current_stream_ptr()))
    return out


def generate_pafs(instances: torch.Tensor, img_hw: Tuple[int, int], sigma: float = 1.5, output_stride: int = 2, edge_inds: Sequence[Tuple[int, int]] = ()) -> torch.Tensor:
    """``instances``: (B, I, N, 2) on the GPU -> (B, 2E, h, w) (the reference's ``flatten_channels=True`` layout)."""
    L.require_cuda(instances, "instances")
    pts = instances.to(torch.float32).contiguous()
    B, I, N, _ = pts.shape
    e = torch.tensor([list(x) for x in edge_inds], dtype=torch.int32, device=pts.device).reshape(-1, 2).contiguous()
    E = int(e.shape[0])
    h, w = _grid(img_hw[0], output_stride), _grid(img_hw[1], output_stride)
    out = torch.empty((B, 2 * E, h, w), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        L.check(L.lib().ph_render_pafs(C.c_void_p(pts.data_ptr()), C.c_void_p(e.data_ptr()), B, I, N, E, int(img_hw[0]), int(img_hw[1]), int(output_stride), float(sigma),
                                       C.c_void_p(out.data_ptr()), L.current_stream_ptr()))
    return out
