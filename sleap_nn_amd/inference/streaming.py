"""GPU-stage -> CPU-stage hand-off types of bottom-up inference.

Mirrors ``sleap_nn/inference/streaming.py:43-320`` (ScoredBatch, GroupingParams,
group_scored_batch).  The reference keeps per-sample Python lists of tensors; here the
batch is stored flattened (one array per field + per-sample offsets) because the CPU stage
is a single C++ call (csrc/group_host.cpp) instead of per-sample/per-edge Python loops.
List-of-tensor views with the reference's field names are provided for API compatibility.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo


@dataclass(eq=False)
class ScoredBatch:
    """Peaks + scored PAF candidates of one batch, host resident (numpy; picklable)."""

    peaks_xy: np.ndarray  # (n_peaks, 2) f32, scaled-input pixel space
    peak_vals: np.ndarray  # (n_peaks,) f32
    peak_channel: np.ndarray  # (n_peaks,) i32
    peak_offsets: np.ndarray  # (B+1,) i32
    cand_edge: np.ndarray  # (n_cand,) i32
    cand_src: np.ndarray  # (n_cand,) i32 sample-local peak index
    cand_dst: np.ndarray
    cand_score: np.ndarray  # (n_cand,) f32
    cand_offsets: np.ndarray  # (B+1,) i32
    info: PreprocInfo
    n_samples: int
    n_nodes: int
    skip_paf: bool = False
    cms: Optional[torch.Tensor] = None
    pafs: Optional[torch.Tensor] = None  # (B, 2E, H, W)

    def _split(self, arr, offs):
        return [torch.from_numpy(np.ascontiguousarray(arr[offs[b] : offs[b + 1]])) for b in range(self.n_samples)]

    # reference-named per-sample list views (streaming.py:76-81)
    @property
    def cms_peaks(self) -> List[torch.Tensor]:
        return self._split(self.peaks_xy, self.peak_offsets)

    @property
    def cms_peak_vals(self) -> List[torch.Tensor]:
        return self._split(self.peak_vals, self.peak_offsets)

    @property
    def cms_peak_channel_inds(self) -> List[torch.Tensor]:
        return self._split(self.peak_channel, self.peak_offsets)

    @property
    def edge_inds(self) -> List[torch.Tensor]:
        return [] if self.skip_paf else self._split(self.cand_edge, self.cand_offsets)

    @property
    def edge_peak_inds(self) -> List[torch.Tensor]:
        if self.skip_paf:
            return []
        pr = np.stack([self.cand_src, self.cand_dst], axis=1) if self.cand_src.size else np.zeros((0, 2), np.int32)
        return self._split(pr, self.cand_offsets)

    @property
    def line_scores(self) -> List[torch.Tensor]:
        return [] if self.skip_paf else self._split(self.cand_score, self.cand_offsets)

    def to_cpu(self) -> "ScoredBatch":
        return self  # already host resident


@dataclass(eq=False)
class GroupingParams:
    paf_scorer_kwargs: dict
    max_instances: Optional[int] = None
    return_confmaps: bool = False
    return_pafs: bool = False
    return_paf_graph: bool = False


def _paf_graph(scored: ScoredBatch):
    pr = np.stack([scored.cand_src, scored.cand_dst], axis=1) if scored.cand_src.size else np.zeros((0, 2), np.int32)
    return (
        torch.from_numpy(scored.peaks_xy.reshape(-1, 2).copy()),
        torch.from_numpy(scored.cand_edge.copy()),
        torch.from_numpy(pr.astype(np.int32)),
        torch.from_numpy(scored.cand_score.copy()),
    )


def _finish(outputs: Outputs, scored: ScoredBatch, params: GroupingParams) -> Outputs:
    if params.return_confmaps and scored.cms is not None:
        outputs.pred_confmaps = scored.cms
    if params.return_pafs and scored.pafs is not None:
        outputs.pred_pafs = scored.pafs
    if params.return_paf_graph:
        outputs.pred_paf_graph = _paf_graph(scored)
    return outputs


def group_scored_batch(scored: ScoredBatch, params: GroupingParams) -> Outputs:
    """CPU grouping stage (streaming.py:147-255): matching + assembly + scale undo + NaN pad."""
    from sleap_nn_amd.inference.ops.paf import PAFScorer, group_batch_host

    B, n_nodes, info = scored.n_samples, scored.n_nodes, scored.info
    if scored.skip_paf:
        mi = params.max_instances or 1
        out = Outputs(
            pred_keypoints=torch.full((B, mi, n_nodes, 2), float("nan")),
            pred_peak_values=torch.full((B, mi, n_nodes), float("nan")),
            instance_scores=torch.full((B, mi), float("nan")),
            preprocess_info=info,
        )
        return _finish(out, scored, params)
    kw = params.paf_scorer_kwargs
    scorer = kw if isinstance(kw, PAFScorer) else PAFScorer(**kw)
    # An instance needs at least one matched edge, so the per-sample instance count is bounded
    # by the number of peaks; group with that bound, then size the output like the reference.
    bound = int(np.max(np.diff(scored.peak_offsets))) if B > 0 else 0
    cap = params.max_instances if params.max_instances is not None else max(1, bound)
    kp, vals, scores, n_inst = group_batch_host(
        n_nodes, scorer.edge_inds, scored.peaks_xy, scored.peak_vals, scored.peak_channel, scored.peak_offsets, scored.cand_edge,
        scored.cand_src, scored.cand_dst, scored.cand_score, scored.cand_offsets, scorer.min_line_scores, scorer.min_instance_peaks,
        max(1, cap), params.max_instances is not None,
    )
    if params.max_instances is None:
        mi = max(1, int(n_inst.max()) if n_inst.size else 1)
        kp, vals, scores = kp[:, :mi], vals[:, :mi], scores[:, :mi]
    kpt = torch.from_numpy(np.ascontiguousarray(kp))
    if info.input_scale != 1.0:
        kpt = kpt / info.input_scale
    eff = info.eff_scale
    if not torch.all(eff == 1.0):
        kpt = kpt / eff.detach().cpu().view(-1, 1, 1, 1)
    out = Outputs(
        pred_keypoints=kpt,
        pred_peak_values=torch.from_numpy(np.ascontiguousarray(vals)),
        instance_scores=torch.from_numpy(np.ascontiguousarray(scores)),
        preprocess_info=info,
    )
    return _finish(out, scored, params)
