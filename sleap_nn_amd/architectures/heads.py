"""Head descriptors (channel count / activation / output key) of the reference.

Mirrors the class names of ``sleap_nn/architectures/heads.py:12-700`` because the head
*class name* is the key of the backend's output dict (architectures/model.py:250-259) and of
the checkpoint (``head_layers.{i}.{HeadName}.0.weight``).  Heads are 1x1 convolutions with
identity activation, except ClassMapsHead (sigmoid) -- heads.py:38-41,58-67,403-405.
"""
from __future__ import annotations

from typing import List, Optional, Sequence


class Head:
    """Base head: ``output_stride`` and ``loss_weight`` (heads.py:12-69)."""

    def __init__(self, output_stride: int = 1, loss_weight: float = 1.0) -> None:
        self.output_stride = int(output_stride)
        self.loss_weight = float(loss_weight)

    @property
    def name(self) -> str:
        return type(self).__name__

    @property
    def channels(self) -> int:
        raise NotImplementedError("Subclasses must implement this method.")

    @property
    def activation(self) -> str:
        return "identity"

    @property
    def loss_function(self) -> str:
        return "mse"


class _PartsHead(Head):
    def __init__(self, part_names: Sequence[str], sigma: float = 5.0, output_stride: int = 1, loss_weight: float = 1.0, **_ignored) -> None:
        super().__init__(output_stride, loss_weight)
        self.part_names = list(part_names)
        self.sigma = sigma

    @property
    def channels(self) -> int:
        return len(self.part_names)


class SingleInstanceConfmapsHead(_PartsHead):
    """heads.py:72-130."""


class CenteredInstanceConfmapsHead(_PartsHead):
    """heads.py:191-254."""

    def __init__(self, part_names, anchor_part: Optional[str] = None, sigma: float = 5.0, output_stride: int = 1, loss_weight: float = 1.0, **_ignored):
        super().__init__(part_names, sigma, output_stride, loss_weight)
        self.anchor_part = anchor_part


class MultiInstanceConfmapsHead(_PartsHead):
    """heads.py:257-315."""


class CentroidConfmapsHead(Head):
    """heads.py:133-188 (one channel)."""

    def __init__(self, anchor_part: Optional[str] = None, sigma: float = 5.0, output_stride: int = 1, loss_weight: float = 1.0, **_ignored):
        super().__init__(output_stride, loss_weight)
        self.anchor_part = anchor_part
        self.sigma = sigma

    @property
    def channels(self) -> int:
        return 1


class PartAffinityFieldsHead(Head):
    """heads.py:318-371 (two channels per edge: x then y)."""

    def __init__(self, edges, sigma: float = 15.0, output_stride: int = 1, loss_weight: float = 1.0, **_ignored):
        super().__init__(output_stride, loss_weight)
        self.edges = [tuple(e) for e in edges]
        self.sigma = sigma

    @property
    def channels(self) -> int:
        return 2 * len(self.edges)


class ClassMapsHead(Head):
    """heads.py:374-431 (sigmoid activation)."""

    def __init__(self, classes, sigma: float = 5.0, output_stride: int = 1, loss_weight: float = 1.0, **_ignored):
        super().__init__(output_stride, loss_weight)
        self.classes = list(classes)
        self.sigma = sigma

    @property
    def channels(self) -> int:
        return len(self.classes)

    @property
    def activation(self) -> str:
        return "sigmoid"


class ClassVectorsHead(Head):
    """heads.py:434-539: global max pool -> (Linear + ReLU) x num_fc_layers -> Linear -> softmax."""

    def __init__(self, classes, num_fc_layers: int = 1, num_fc_units: int = 64, global_pool: bool = True, output_stride: int = 1, loss_weight: float = 1.0, **_ignored):
        super().__init__(output_stride, loss_weight)
        self.classes = list(classes)
        self.num_fc_layers = int(num_fc_layers)
        self.num_fc_units = int(num_fc_units)
        self.global_pool = bool(global_pool)

    @property
    def channels(self) -> int:
        return len(self.classes)

    @property
    def activation(self) -> str:
        return "softmax"

    @property
    def loss_function(self) -> str:
        return "categorical_crossentropy"


def get_head(model_type: str, head_config) -> List[Head]:
    """Head list per model type, in the reference's order (architectures/model.py:70-154)."""
    from sleap_nn_amd.utils import cfg_get, to_plain

    def kw(key):
        d = to_plain(cfg_get(head_config, key))
        if d is None:
            raise ValueError(f"head config for '{model_type}' is missing '{key}'")
        return d

    if model_type == "single_instance":
        return [SingleInstanceConfmapsHead(**kw("confmaps"))]
    if model_type == "centered_instance":
        return [CenteredInstanceConfmapsHead(**kw("confmaps"))]
    if model_type == "centroid":
        return [CentroidConfmapsHead(**kw("confmaps"))]
    if model_type == "bottomup":
        return [MultiInstanceConfmapsHead(**kw("confmaps")), PartAffinityFieldsHead(**kw("pafs"))]
    if model_type == "multi_class_bottomup":
        return [MultiInstanceConfmapsHead(**kw("confmaps")), ClassMapsHead(**kw("class_maps"))]
    if model_type == "multi_class_topdown":
        return [CenteredInstanceConfmapsHead(**kw("confmaps")), ClassVectorsHead(**kw("class_vectors"))]
    raise Exception(
        f"{model_type} is not a model type of the MI355X hot path. Supported: `single_instance`, "
        "`centered_instance`, `centroid`, `bottomup`, `multi_class_bottomup`, `multi_class_topdown`."
    )
