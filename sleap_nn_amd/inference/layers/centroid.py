"""``CentroidLayer`` (sleap_nn/inference/layers/centroid.py:44-271): centroid confmap -> local
peaks -> per-frame top-k, NaN-padded ``(B, I, 2)``."""
from __future__ import annotations

from typing import Optional

import torch

from sleap_nn_amd.inference.backends import ModelBackend
from sleap_nn_amd.inference.layers.base import InferenceLayer
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig
from sleap_nn_amd.inference.ops.coord import undo_eff_scale, undo_input_scale, undo_stride
from sleap_nn_amd.inference.ops.peaks import find_local_peaks
from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo


class CentroidLayer(InferenceLayer):
    _HEAD_OUTPUT_KEY = "CentroidConfmapsHead"

    def __init__(self, backend: ModelBackend, output_stride: int, max_instances: Optional[int] = None, max_stride: int = 1,
                 anchor_ind: Optional[int] = None, use_gt_centroids: bool = False, preprocess_config: Optional[PreprocessConfig] = None,
                 postprocess_config: Optional[PostprocessConfig] = None) -> None:
        super().__init__(backend, preprocess_config or PreprocessConfig(), postprocess_config or PostprocessConfig(max_instances=max_instances), output_stride, max_stride)
        if use_gt_centroids:
            raise NotImplementedError("use_gt_centroids (LabelsReader path) is outside the MI355X hot path")
        self.max_instances = max_instances
        self.anchor_ind = anchor_ind
        self.use_gt_centroids = False

    def postprocess(self, raw_out: dict, info: PreprocInfo) -> Outputs:
        cms = self._extract_confmaps(raw_out)
        pc = self.postprocess_config
        peaks, vals, sb, _ = find_local_peaks(cms, threshold=pc.peak_threshold, refinement=pc.effective_refinement, integral_patch_size=pc.integral_patch_size)
        peaks = undo_input_scale(undo_stride(peaks, info.output_stride), info.input_scale)
        B = int(cms.shape[0])
        counts = torch.bincount(sb.long(), minlength=B) if sb.numel() else torch.zeros(B, dtype=torch.long, device=cms.device)
        max_instances = getattr(pc, "max_instances", None) or self.max_instances or (int(counts.max().item()) if sb.numel() else 0)
        if max_instances == 0:
            max_instances = 1
        dev = peaks.device
        cp = torch.full((B, max_instances, 2), float("nan"), device=dev)
        cv = torch.full((B, max_instances), float("nan"), device=dev)
        offs = torch.cumsum(counts, 0) - counts
        counts_h, offs_h = counts.tolist(), offs.tolist()
        for b in range(B):  # peaks are already grouped by sample (reference order)
            n, o = counts_h[b], offs_h[b]
            if n == 0:
                continue
            p, v = peaks[o : o + n], vals[o : o + n]
            if n > max_instances:
                v, idx = torch.topk(v, max_instances)
                p = p[idx]
                n = max_instances
            cp[b, :n], cv[b, :n] = p, v
        cp = undo_eff_scale(cp, info.eff_scale)
        out = Outputs(pred_centroids=cp, pred_centroid_values=cv, preprocess_info=info)
        if pc.return_confmaps:
            out.pred_confmaps = cms.detach()
        return out
