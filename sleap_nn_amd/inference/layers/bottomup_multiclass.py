"""``BottomUpMultiClassLayer`` (sleap_nn/inference/layers/bottomup_multiclass.py:26-190)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from sleap_nn_amd.inference.backends import ModelBackend
from sleap_nn_amd.inference.layers.base import InferenceLayer
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig
from sleap_nn_amd.inference.ops.identity import classify_peaks_from_maps
from sleap_nn_amd.inference.ops.peaks import find_local_peaks
from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo


class BottomUpMultiClassLayer(InferenceLayer):
    def __init__(self, backend: ModelBackend, cms_output_stride: int, class_maps_output_stride: int, max_instances: Optional[int] = None,
                 max_stride: int = 1, preprocess_config: Optional[PreprocessConfig] = None, postprocess_config: Optional[PostprocessConfig] = None) -> None:
        super().__init__(backend, preprocess_config or PreprocessConfig(), postprocess_config or PostprocessConfig(), cms_output_stride, max_stride)
        self.cms_output_stride = cms_output_stride
        self.class_maps_output_stride = class_maps_output_stride
        self.max_instances = max_instances

    def postprocess(self, raw_out: dict, info: PreprocInfo) -> Outputs:
        cms = raw_out["MultiInstanceConfmapsHead"]
        class_maps = raw_out["ClassMapsHead"]
        pc = self.postprocess_config
        peaks, vals, sb, sc = find_local_peaks(cms, threshold=pc.peak_threshold, refinement=pc.effective_refinement, integral_patch_size=pc.integral_patch_size)
        peaks = peaks * self.cms_output_stride
        inst, pvals, cprobs = classify_peaks_from_maps(class_maps, peaks / self.class_maps_output_stride, vals, sb, sc, n_channels=cms.shape[1])
        inst = inst * self.class_maps_output_stride
        if info.input_scale != 1.0:
            inst = inst / info.input_scale
        eff = info.eff_scale.cpu()
        if not torch.all(eff == 1.0):
            inst = inst / eff.view(-1, 1, 1, 1)
        iscores = torch.nanmean(pvals, dim=-1)
        tscores = torch.nanmean(cprobs, dim=-1)
        max_instances = getattr(pc, "max_instances", None) or self.max_instances
        if max_instances is not None:
            for b in range(inst.shape[0]):  # keep the top-N classes by score, in place (bottomup_multiclass.py:152-190)
                s = iscores[b]
                if int((~torch.isnan(s)).sum()) <= max_instances:
                    continue
                order = np.argsort(s.numpy())[::-1]
                drop = torch.ones(s.shape[0], dtype=torch.bool)
                drop[torch.as_tensor(order[:max_instances].copy(), dtype=torch.long)] = False
                inst[b, drop], pvals[b, drop], iscores[b, drop], tscores[b, drop] = float("nan"), float("nan"), float("nan"), float("nan")
        out = Outputs(pred_keypoints=inst, pred_peak_values=pvals, instance_scores=iscores, preprocess_info=info)
        out.instance_tracking_scores = tscores
        if pc.return_confmaps:
            out.pred_confmaps = cms.detach()
        if pc.return_class_maps:
            out.pred_class_maps = class_maps.detach()
        return out
