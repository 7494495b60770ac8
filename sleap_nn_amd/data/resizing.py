"""Resizes of the preprocessing chain on the GPU (sleap_nn/data/resizing.py:10-175).

``tvf.resize`` (bilinear, antialias) is ``ph_resize_bilinear_aa``; padding stays ``F.pad`` (a device memcpy).  Frames are moved
to the device first: everything after this point of the chain is device-resident anyway.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from sleap_nn_amd import _lib as L


def find_padding_for_stride(image_height: int, image_width: int, max_stride: int) -> Tuple[int, int]:
    return (max_stride - (image_height % max_stride)) % max_stride, (max_stride - (image_width % max_stride)) % max_stride


def apply_pad_to_stride(image: torch.Tensor, max_stride: int) -> torch.Tensor:
    """Zero pad bottom/right to a multiple of ``max_stride`` (resizing.py:35-67)."""
    if max_stride > 1:
        ph, pw = find_padding_for_stride(image.shape[-2], image.shape[-1], max_stride)
        if ph > 0 or pw > 0:
            image = F.pad(image, (0, pw, 0, ph), mode="constant")
    return image


def resize_bilinear_aa(image: torch.Tensor, size, device: Optional[torch.device] = None) -> torch.Tensor:
    """``tvf.resize(image, size)`` for uint8 / float32 ``(..., H, W)`` tensors; returns a device tensor."""
    oh, ow = int(size[0]), int(size[1])
    if device is None:
        device = image.device if image.is_cuda else torch.device("cuda", torch.cuda.current_device())
    x = image.to(device)
    if x.dtype not in (torch.uint8, torch.float32):
        x = x.float()
    x = x.contiguous()
    H, W = x.shape[-2:]
    if (H, W) == (oh, ow):
        return x
    planes = x.numel() // (H * W)
    out = torch.empty(x.shape[:-2] + (oh, ow), dtype=x.dtype, device=device)
    tmp = torch.empty(planes * H * ow, dtype=x.dtype, device=device) if (oh != H and ow != W) else None
    stream = torch.cuda.current_stream(device).cuda_stream
    L.check(L.lib().ph_resize_bilinear_aa(C.c_void_p(x.data_ptr()), 0 if x.dtype == torch.uint8 else 1, planes, H, W, C.c_void_p(out.data_ptr()), oh, ow,
                                          C.c_void_p(tmp.data_ptr() if tmp is not None else 0), C.c_void_p(stream)))
    return out


def resize_image(image: torch.Tensor, scale: float) -> torch.Tensor:
    """Rescale by ``scale`` (resizing.py:70-84: ``new_size = [int(H * scale), int(W * scale)]``)."""
    h, w = image.shape[-2:]
    return resize_bilinear_aa(image, [int(h * scale), int(w * scale)])


def apply_sizematcher(image: torch.Tensor, max_height: Optional[int] = None, max_width: Optional[int] = None):
    """Fit a (C, H, W) frame into (max_height, max_width) keeping its aspect ratio, zero pad bottom/right; returns
    ``(image, eff_scale)`` (resizing.py:136-175)."""
    h, w = image.shape[-2:]
    max_height = h if max_height is None else max_height
    max_width = w if max_width is None else max_width
    if h == max_height and w == max_width:
        return image, 1.0
    hratio, wratio = max_height / h, max_width / w
    if hratio > wratio:
        eff, th, tw = wratio, int(round(h * wratio)), int(round(w * wratio))
    else:
        eff, tw, th = hratio, int(round(w * hratio)), int(round(h * hratio))
    image = resize_bilinear_aa(image, (th, tw))
    image = F.pad(image, (0, max_width - tw, 0, max_height - th), mode="constant")
    return image, eff
