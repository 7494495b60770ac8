"""Peak finding on the GPU -- same call signatures and return conventions as
``sleap_nn/inference/ops/peaks.py`` (find_local_peaks :221-259, find_global_peaks :133-181),
implemented by the wavefront local-maxima / argmax kernels of csrc/post_kernels.hip."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from sleap_nn_amd import _lib as L


def _prep(cms: torch.Tensor) -> torch.Tensor:
    if cms.dim() != 4:
        raise ValueError(f"cms must be (samples, channels, height, width), got {tuple(cms.shape)}")
    L.require_cuda(cms, "cms")
    return cms.detach().to(torch.float32).contiguous()


def _refine_flag(refinement: Optional[str]) -> int:
    return 1 if refinement == "integral" else 0


def find_local_peaks_device(cms: torch.Tensor, threshold: float = 0.2, refinement: Optional[str] = None, integral_patch_size: int = 5, capacity: Optional[int] = None,
                            xy_scale: float = 1.0, out=None):
    """Device-resident result: (xy (cap,2), vals (cap,), sample (cap,), channel (cap,), counts int32[2+2B]).

    ``counts[0]`` = total number of peaks, ``counts[1+b]`` = peaks of sample ``b``,
    ``counts[1+B:2+2B]`` = exclusive per-sample offsets (B+1 entries); no host
    sync happens here.  Rows beyond ``counts[0]`` are undefined.
    ``xy_scale``: the kernel multiplies the (refined) coordinates by it (``peaks * cms_output_stride`` of the bottom-up
    layer, bottomup.py:111).  ``out``: optional pre-allocated ``(xy, vals, channel, counts)`` views (a caller's packed
    D2H arena) the kernel writes directly.
    """
    cms = _prep(cms)
    B, Cc, H, W = cms.shape
    dev = cms.device
    cap = int(capacity) if capacity is not None else max(1024, B * Cc * 64)
    if out is not None:
        xy, vals, sc, counts = out
        assert xy.numel() == 2 * cap and vals.numel() == cap and sc.numel() == cap and counts.numel() == 2 + 2 * B
    else:
        xy = torch.empty((cap, 2), dtype=torch.float32, device=dev)
        vals = torch.empty((cap,), dtype=torch.float32, device=dev)
        sc = torch.empty((cap,), dtype=torch.int32, device=dev)
        counts = torch.empty((2 + 2 * B,), dtype=torch.int32, device=dev)
    sb = torch.empty((cap,), dtype=torch.int32, device=dev)
    scratch = torch.empty((int(L.lib().ph_local_peaks_scratch_bytes(B, Cc, H, W)) // 4,), dtype=torch.int32, device=dev)  # sized for the one-pass kernels
    with torch.cuda.device(dev):
        L.check(
            L.lib().ph_local_peaks(
                C.c_void_p(cms.data_ptr()), B, Cc, H, W, float(threshold), _refine_flag(refinement), int(integral_patch_size),
                C.c_void_p(xy.data_ptr()), C.c_void_p(vals.data_ptr()), C.c_void_p(sb.data_ptr()), C.c_void_p(sc.data_ptr()),
                C.c_void_p(counts.data_ptr()), cap, float(xy_scale), C.c_void_p(scratch.data_ptr()), scratch.numel() * 4, L.current_stream_ptr(),
            )
        )
    return xy, vals, sb, sc, counts, cms


def find_local_peaks(cms: torch.Tensor, threshold: float = 0.2, refinement: Optional[str] = None, integral_patch_size: int = 5) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """``(peak_points (n,2) xy, peak_vals (n,), peak_sample_inds (n,) i32, peak_channel_inds (n,) i32)``
    in the reference's (sample, y, x, channel) order."""
    cap = None
    while True:
        xy, vals, sb, sc, counts, _ = find_local_peaks_device(cms, threshold, refinement, integral_patch_size, cap)
        n = int(counts[0].item())  # the same sync point torch.where has in the reference
        if n <= xy.shape[0]:
            return xy[:n], vals[:n], sb[:n], sc[:n]
        cap = n


def find_local_peaks_rough(cms: torch.Tensor, threshold: float = 0.2):
    return find_local_peaks(cms, threshold, None)


def find_global_peaks(cms: torch.Tensor, threshold: float = 0.2, refinement: Optional[str] = None, integral_patch_size: int = 5) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(peak_points (B,C,2) xy with NaN below threshold, peak_vals (B,C))``."""
    cms = _prep(cms)
    B, Cc, H, W = cms.shape
    xy = torch.empty((B, Cc, 2), dtype=torch.float32, device=cms.device)
    vals = torch.empty((B, Cc), dtype=torch.float32, device=cms.device)
    with torch.cuda.device(cms.device):
        L.check(
            L.lib().ph_global_peaks(
                C.c_void_p(cms.data_ptr()), B, Cc, H, W, float(threshold), _refine_flag(refinement), int(integral_patch_size),
                C.c_void_p(xy.data_ptr()), C.c_void_p(vals.data_ptr()), L.current_stream_ptr(),
            )
        )
    return xy, vals


def find_global_peaks_rough(cms: torch.Tensor, threshold: float = 0.1):
    return find_global_peaks(cms, threshold, None)
