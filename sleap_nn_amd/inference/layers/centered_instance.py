"""``CenteredInstanceLayer`` (sleap_nn/inference/layers/centered_instance.py:39-230): crops ->
centered-instance confmaps -> one global peak per node (crop-local coordinates)."""
from __future__ import annotations

from typing import Optional

import torch

from sleap_nn_amd.inference.backends import ModelBackend
from sleap_nn_amd.inference.layers.base import InferenceLayer
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig
from sleap_nn_amd.inference.ops.coord import undo_stride
from sleap_nn_amd.inference.ops.peaks import find_global_peaks
from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo


class CenteredInstanceLayer(InferenceLayer):
    _HEAD_OUTPUT_KEY = "CenteredInstanceConfmapsHead"

    def __init__(self, backend: ModelBackend, output_stride: int, max_stride: int = 1, use_gt_peaks: bool = False,
                 preprocess_config: Optional[PreprocessConfig] = None, postprocess_config: Optional[PostprocessConfig] = None) -> None:
        super().__init__(backend, preprocess_config or PreprocessConfig(), postprocess_config or PostprocessConfig(), output_stride, max_stride)
        if use_gt_peaks:
            raise NotImplementedError("use_gt_peaks (LabelsReader path) is outside the MI355X hot path")
        self.use_gt_peaks = False

    def preprocess(self, image):
        """Crops arrive model-sized: only the stride pad + n_samples axis apply (no sizematcher)."""
        x = self._to_4d_tensor(image)
        from sleap_nn_amd.inference.layers.base import apply_pad_to_stride

        H, W = x.shape[-2:]
        x = apply_pad_to_stride(x, self.max_stride) if self.max_stride != 1 else x
        info = PreprocInfo(original_size=(H, W), processed_size=tuple(x.shape[-2:]), eff_scale=torch.ones(x.shape[0]), input_scale=1.0, output_stride=self.output_stride)
        return x.unsqueeze(1), info

    def postprocess(self, raw_out: dict, info: PreprocInfo) -> Outputs:
        cms = self._extract_confmaps(raw_out)
        pc = self.postprocess_config
        peaks, vals = find_global_peaks(cms, threshold=pc.peak_threshold, refinement=pc.effective_refinement, integral_patch_size=pc.integral_patch_size)
        peaks = undo_stride(peaks, info.output_stride)
        out = Outputs(pred_keypoints=peaks.unsqueeze(1), pred_crop_keypoints=peaks.unsqueeze(1), pred_peak_values=vals.unsqueeze(1), preprocess_info=info)
        if pc.return_confmaps:
            out.pred_confmaps = cms.detach()
        return out
