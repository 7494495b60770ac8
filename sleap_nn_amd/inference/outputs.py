"""``Outputs`` container (subset of sleap_nn/inference/outputs.py:64-779 that the hot path fills)."""
from __future__ import annotations

from dataclasses import dataclass, fields, replace
from typing import Any, Dict, Optional, Tuple

import torch

from sleap_nn_amd.inference.preprocess_info import PreprocInfo

_HEAVY = ("original_image", "processed_image", "crops", "pred_confmaps", "pred_pafs", "pred_class_maps", "pred_paf_graph")


@dataclass(eq=False, repr=False)
class Outputs:
    """Shape convention: B batch, I max instances, N nodes; NaN = missing (outputs.py:64-72)."""

    original_image: Optional[torch.Tensor] = None
    processed_image: Optional[torch.Tensor] = None
    crops: Optional[torch.Tensor] = None
    pred_keypoints: Optional[torch.Tensor] = None  # (B, I, N, 2) image (x, y)
    pred_crop_keypoints: Optional[torch.Tensor] = None
    pred_peak_values: Optional[torch.Tensor] = None  # (B, I, N)
    pred_confmaps: Optional[torch.Tensor] = None  # (B, N, H, W)
    pred_pafs: Optional[torch.Tensor] = None  # (B, 2E, H, W)
    pred_centroids: Optional[torch.Tensor] = None
    pred_centroid_values: Optional[torch.Tensor] = None
    instance_scores: Optional[torch.Tensor] = None  # (B, I)
    instance_valid: Optional[torch.Tensor] = None
    instance_bboxes: Optional[torch.Tensor] = None  # (B, I, 4, 2)
    instance_tracking_scores: Optional[torch.Tensor] = None  # (B, I)
    pred_class_maps: Optional[torch.Tensor] = None
    pred_class_inds: Optional[torch.Tensor] = None  # (B, I, N) int64, -1 = unassigned
    pred_class_probs: Optional[torch.Tensor] = None  # stage 2 of multi-class top-down: (n_crops, 1, n_classes)
    pred_class_vectors: Optional[torch.Tensor] = None  # (B, I, n_classes) when requested
    pred_paf_graph: Optional[Tuple[torch.Tensor, ...]] = None
    preprocess_info: Optional[PreprocInfo] = None
    frame_indices: Optional[torch.Tensor] = None
    video_indices: Optional[torch.Tensor] = None

    def _map(self, fn) -> "Outputs":
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, torch.Tensor):
                kw[f.name] = fn(v)
            elif isinstance(v, tuple) and v and isinstance(v[0], torch.Tensor):
                kw[f.name] = tuple(fn(t) for t in v)
            else:
                kw[f.name] = v
        return Outputs(**kw)

    def to(self, device) -> "Outputs":
        return self._map(lambda t: t.to(device))

    def cpu(self) -> "Outputs":
        return self.to("cpu")

    def detach(self) -> "Outputs":
        return self._map(lambda t: t.detach())

    def numpy(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                out[f.name] = v.detach().cpu().numpy()
            elif isinstance(v, tuple) and v and isinstance(v[0], torch.Tensor):
                out[f.name] = tuple(t.detach().cpu().numpy() for t in v)
            elif isinstance(v, PreprocInfo):
                out[f.name] = v.cpu()
            else:
                out[f.name] = v
        return out

    def slim(self) -> "Outputs":
        """Drop heavy intermediates, detach + CPU: pickle-safe (outputs.py slim contract)."""
        o = replace(self, **{k: None for k in _HEAVY})
        o = o._map(lambda t: t.detach().cpu())
        if o.preprocess_info is not None:
            o = replace(o, preprocess_info=o.preprocess_info.cpu())
        return o

    def __repr__(self) -> str:
        parts = []
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, torch.Tensor):
                parts.append(f"{f.name}=Tensor{tuple(v.shape)}")
        return f"Outputs({', '.join(parts)})" if parts else "Outputs(empty)"
