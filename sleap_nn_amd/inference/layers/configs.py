"""Pre/post-process knobs (sleap_nn/inference/layers/configs.py:19-98)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple


@dataclass(frozen=True)
class PreprocessConfig:
    ensure_rgb: Optional[bool] = None
    ensure_grayscale: Optional[bool] = None
    max_height: Optional[int] = None
    max_width: Optional[int] = None
    scale: float = 1.0
    crop_size: Optional[Tuple[int, int]] = None

    def __post_init__(self) -> None:
        if self.ensure_rgb and self.ensure_grayscale:
            raise ValueError(
                "ensure_rgb and ensure_grayscale cannot both be True; choose one "
                "(or leave both None to keep the source channel count)."
            )


@dataclass(frozen=True)
class PostprocessConfig:
    peak_threshold: float = 0.2
    refinement: str = "integral"  # "integral" | "none"
    integral_patch_size: int = 5
    max_instances: Optional[int] = None
    return_confmaps: bool = False
    return_pafs: bool = False
    return_paf_graph: bool = False
    return_class_maps: bool = False
    return_class_vectors: bool = False

    @property
    def effective_refinement(self) -> Optional[str]:
        return self.refinement if self.refinement != "none" else None
