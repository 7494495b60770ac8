"""Coordinate ladder (sleap_nn/inference/ops/coord.py:27-90) -- host-side scalar plumbing."""
from __future__ import annotations

import torch


def undo_stride(coords: torch.Tensor, output_stride: int) -> torch.Tensor:
    return coords if output_stride == 1 else coords * output_stride


def undo_input_scale(coords: torch.Tensor, input_scale: float) -> torch.Tensor:
    return coords if input_scale == 1.0 else coords / input_scale


def undo_eff_scale(coords: torch.Tensor, eff_scale: torch.Tensor) -> torch.Tensor:
    if torch.all(eff_scale == 1.0):
        return coords
    shape = [eff_scale.shape[0]] + [1] * (coords.ndim - 1)
    return coords / eff_scale.view(shape).to(coords.device)


def add_crop_offset(peaks: torch.Tensor, crop_topleft: torch.Tensor) -> torch.Tensor:
    return peaks + crop_topleft.to(peaks.device).view(-1, 1, 2)
