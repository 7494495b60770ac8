"""Bounding boxes + crop gather (sleap_nn/inference/ops/crops.py:31-124,
sleap_nn/data/instance_cropping.py:129-171); the gather is the ``ph_crop_bboxes`` kernel."""
from __future__ import annotations

import ctypes as C

import torch

from sleap_nn_amd import _lib as L


def make_centered_bboxes(centroids: torch.Tensor, box_height: int, box_width: int) -> torch.Tensor:
    """(n, 4, 2) corners TL, TR, BR, BL of a ``box_height x box_width`` box centred on each (x, y)."""
    hw, hh = box_width / 2, box_height / 2
    x, y = centroids[..., 0], centroids[..., 1]
    corners = torch.stack(
        [torch.stack([x - hw, y - hh], -1), torch.stack([x + hw, y - hh], -1), torch.stack([x + hw, y + hh], -1), torch.stack([x - hw, y + hh], -1)], dim=-2
    )
    off = torch.tensor([[0.5, 0.5], [-0.5, 0.5], [-0.5, -0.5], [0.5, -0.5]], device=corners.device)
    return corners + off


def crop_bboxes(images: torch.Tensor, bboxes: torch.Tensor, sample_inds: torch.Tensor) -> torch.Tensor:
    """``(n, C, h, w)`` zero-padded crops of ``images`` (B, C, H, W; uint8 or float32) on the GPU."""
    L.require_cuda(images, "images")
    n = int(bboxes.shape[0])
    if n == 0:
        return torch.empty(0, images.shape[1], 0, 0, device=images.device, dtype=images.dtype)
    bb = bboxes.detach().to("cpu", torch.float32)
    h = int(abs(bb[0, 3, 1] - bb[0, 0, 1]).item()) + 1
    w = int(abs(bb[0, 1, 0] - bb[0, 0, 0]).item()) + 1
    if images.dtype == torch.uint8:
        code = 0
    elif images.dtype == torch.float32:
        code = 1
    else:
        raise TypeError(f"crop_bboxes supports uint8 and float32 images, got {images.dtype}")
    images = images.contiguous()
    B, Cc, H, W = images.shape
    dev = images.device
    tl = bboxes[:, 0, :].detach().to(dev, torch.float32).contiguous()
    si = torch.as_tensor(sample_inds).to(dev, torch.int32).contiguous()
    out = torch.empty((n, Cc, h, w), dtype=images.dtype, device=dev)
    with torch.cuda.device(dev):
        L.check(
            L.lib().ph_crop_bboxes(
                C.c_void_p(images.data_ptr()), code, B, Cc, H, W, C.c_void_p(tl.data_ptr()), C.c_void_p(si.data_ptr()), n, h, w,
                C.c_void_p(out.data_ptr()), L.current_stream_ptr(),
            )
        )
    return out
