"""Multi-class top-down (sleap_nn/inference/layers/topdown_multiclass.py:27-185 and the class
scatter of layers/topdown.py:333-390): the centered-instance model also emits a ``ClassVectorsHead``
softmax per crop; classes are assigned by Hungarian matching PER FRAME (crops of different frames
never compete for a class slot), the class probability rides in ``instance_tracking_scores``."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from sleap_nn_amd.inference.backends import ModelBackend
from sleap_nn_amd.inference.layers.centered_instance import CenteredInstanceLayer
from sleap_nn_amd.inference.layers.centroid import CentroidLayer
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig
from sleap_nn_amd.inference.layers.topdown import TopDownLayer
from sleap_nn_amd.inference.ops.identity import get_class_inds_from_vectors
from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo


class CenteredInstanceMultiClassLayer(CenteredInstanceLayer):
    """Per crop: keypoints as ``CenteredInstanceLayer`` + the class-probability vector."""

    def __init__(self, backend: ModelBackend, output_stride: int, max_stride: int = 1, preprocess_config: Optional[PreprocessConfig] = None,
                 postprocess_config: Optional[PostprocessConfig] = None, class_names: Optional[list] = None) -> None:
        super().__init__(backend, output_stride, max_stride, False, preprocess_config, postprocess_config)
        self.class_names = list(class_names) if class_names is not None else None

    def postprocess(self, raw_out: dict, info: PreprocInfo) -> Outputs:
        out = super().postprocess(raw_out, info)
        if "ClassVectorsHead" not in raw_out:
            raise KeyError(f"backend output has no 'ClassVectorsHead' (keys: {sorted(raw_out)})")
        probs = raw_out["ClassVectorsHead"].detach()  # (n_crops, n_classes)
        inds, pr = get_class_inds_from_vectors(probs)  # joint over the batch, as the reference's inner layer does
        n_nodes = out.pred_keypoints.shape[-2]
        out.pred_class_inds = inds.view(-1, 1, 1).expand(-1, 1, n_nodes)
        out.pred_class_probs = probs.unsqueeze(1)
        out.instance_tracking_scores = pr.view(-1, 1)
        if getattr(self.postprocess_config, "return_class_vectors", False):
            out.pred_class_vectors = probs
        return out


class TopDownMultiClassLayer(TopDownLayer):
    def __init__(self, centroid_layer: CentroidLayer, centered_instance_layer: CenteredInstanceMultiClassLayer, crop_size: Tuple[int, int],
                 centroid_nms: bool = False, centroid_nms_threshold: float = 0.5, return_crops: bool = False) -> None:
        if not isinstance(centered_instance_layer, CenteredInstanceMultiClassLayer):
            raise TypeError("TopDownMultiClassLayer requires a CenteredInstanceMultiClassLayer for the centered_instance_layer argument; got "
                            f"{type(centered_instance_layer).__name__}.")
        super().__init__(centroid_layer, centered_instance_layer, crop_size, centroid_nms, centroid_nms_threshold, return_crops)

    @property
    def class_names(self):
        return self.centered_instance_layer.class_names
