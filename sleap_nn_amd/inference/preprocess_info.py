"""``PreprocInfo`` value type (sleap_nn/inference/preprocess_info.py:20-84)."""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Optional, Tuple

import torch


@dataclass(frozen=True, eq=False)
class PreprocInfo:
    original_size: Tuple[int, int] = (0, 0)
    processed_size: Tuple[int, int] = (0, 0)
    eff_scale: torch.Tensor = field(default_factory=lambda: torch.tensor([1.0]))
    input_scale: float = 1.0
    output_stride: int = 1
    pad_amount: Tuple[int, int] = (0, 0)
    crop_offsets: Optional[torch.Tensor] = None

    def cpu(self) -> "PreprocInfo":
        return replace(
            self,
            eff_scale=self.eff_scale.detach().cpu(),
            crop_offsets=self.crop_offsets.detach().cpu() if self.crop_offsets is not None else None,
        )

    def __repr__(self) -> str:
        return (
            f"PreprocInfo(orig={self.original_size}, proc={self.processed_size}, eff_scale=Tensor{tuple(self.eff_scale.shape)}, "
            f"input_scale={self.input_scale}, output_stride={self.output_stride}, pad={self.pad_amount})"
        )
