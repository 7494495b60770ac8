"""``SingleInstanceLayer`` (sleap_nn/inference/layers/single_instance.py:36-108)."""
from __future__ import annotations

from typing import Optional

from sleap_nn_amd.inference.backends import ModelBackend
from sleap_nn_amd.inference.layers.base import InferenceLayer
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig
from sleap_nn_amd.inference.ops.coord import undo_eff_scale, undo_input_scale, undo_stride
from sleap_nn_amd.inference.ops.peaks import find_global_peaks
from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo


class SingleInstanceLayer(InferenceLayer):
    _HEAD_OUTPUT_KEY = "SingleInstanceConfmapsHead"
    _GRAPHABLE_POSTPROCESS = True  # global peaks + refinement + the coordinate ladder: all device work (``predict_graphed``)

    def __init__(self, backend: ModelBackend, output_stride: int, max_stride: int = 1, preprocess_config: Optional[PreprocessConfig] = None, postprocess_config: Optional[PostprocessConfig] = None) -> None:
        super().__init__(backend, preprocess_config or PreprocessConfig(), postprocess_config or PostprocessConfig(), output_stride, max_stride)

    def postprocess(self, raw_out: dict, info: PreprocInfo) -> Outputs:
        cms = self._extract_confmaps(raw_out)
        pc = self.postprocess_config
        peaks, vals = find_global_peaks(cms, threshold=pc.peak_threshold, refinement=pc.effective_refinement, integral_patch_size=pc.integral_patch_size)
        peaks = undo_stride(peaks, info.output_stride)
        peaks = undo_input_scale(peaks, info.input_scale)
        peaks = undo_eff_scale(peaks, info.eff_scale)
        out = Outputs(pred_keypoints=peaks.unsqueeze(1), pred_peak_values=vals.unsqueeze(1), preprocess_info=info)
        if pc.return_confmaps:
            out.pred_confmaps = cms.detach()
        return out
