"""Coordinate ladder (sleap_nn/inference/ops/coord.py:27-90) -- host-side scalar plumbing."""
from __future__ import annotations

import torch


def undo_stride(coords: torch.Tensor, output_stride: int) -> torch.Tensor:
    return coords if output_stride == 1 else coords * output_stride


def undo_input_scale(coords: torch.Tensor, input_scale: float) -> torch.Tensor:
    return coords if input_scale == 1.0 else coords / input_scale


_EFF_ON_DEVICE = {}


def undo_eff_scale(coords: torch.Tensor, eff_scale: torch.Tensor) -> torch.Tensor:
    if torch.all(eff_scale == 1.0):
        return coords
    shape = [eff_scale.shape[0]] + [1] * (coords.ndim - 1)
    if eff_scale.device != coords.device and eff_scale.device.type == "cpu":
        # the per-frame scales of a batch are a handful of host floats: their device copy is kept by value, so that a step captured in a hipGraph (predict_graphed: the warm-up
        # run outside the capture makes the copy) finds it there instead of issuing a host-to-device copy inside the capture
        key = (tuple(eff_scale.flatten().tolist()), str(coords.device))
        dev = _EFF_ON_DEVICE.get(key)
        if dev is None:
            if len(_EFF_ON_DEVICE) > 256:
                _EFF_ON_DEVICE.clear()
            dev = _EFF_ON_DEVICE[key] = eff_scale.to(coords.device)
        return coords / dev.view(shape)
    return coords / eff_scale.view(shape).to(coords.device)


def add_crop_offset(peaks: torch.Tensor, crop_topleft: torch.Tensor) -> torch.Tensor:
    return peaks + crop_topleft.to(peaks.device).view(-1, 1, 2)
