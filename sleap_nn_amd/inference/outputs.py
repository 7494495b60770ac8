"""``Outputs`` container (subset of sleap_nn/inference/outputs.py:64-779 that the hot path fills) and its packaging into
``sleap_io`` objects (``to_instances`` :284-432, centroid-only packaging :434-477, ``to_labels`` :612-779 restricted to the pose
fields; ``sleap_io`` is imported lazily, only by those methods)."""
from __future__ import annotations

from dataclasses import dataclass, fields, replace
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np
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

    # -- packaging into sleap_io objects (outputs.py:284-477, 612-779) -------------------------------------------
    @property
    def batch_size(self) -> int:
        for name in ("pred_keypoints", "pred_centroids", "pred_peak_values", "instance_scores"):
            v = getattr(self, name)
            if v is not None:
                return int(v.shape[0])
        return 0

    def _host(self, name: str, b: int) -> Optional[np.ndarray]:
        v = getattr(self, name)
        return None if v is None else v[b].detach().cpu().numpy()

    def _instance_rows(self, b: int, n_nodes: int, anchor_ind: Optional[int], n_tracks: Optional[int]) -> Iterator[Dict[str, Any]]:
        """One dict per non-empty instance slot of sample ``b``: ``points (N, 2)``, ``point_scores (N,)``, ``score``, and for the
        multi-class paths ``track_index`` / ``tracking_score``."""
        kp = self._host("pred_keypoints", b)
        if kp is None:
            cen = self._host("pred_centroids", b)
            if cen is None:
                return
            # centroid-only packaging (:434-477): the centroid sits at `anchor_ind` (node 0 if unset), every other node is NaN;
            # score = centroid value (0 when that is NaN)
            a = 0 if anchor_ind is None else int(anchor_ind)
            if not 0 <= a < n_nodes:
                raise ValueError(f"anchor_ind={a} is out of range for skeleton with {n_nodes} nodes.")
            cv = self._host("pred_centroid_values", b)
            for i in range(cen.shape[0]):
                if np.isnan(cen[i]).any():
                    continue
                pts = np.full((n_nodes, 2), np.nan, dtype=np.float32)
                sc = np.full((n_nodes,), np.nan, dtype=np.float32)
                v = float(cv[i]) if cv is not None else float("nan")
                pts[a], sc[a] = cen[i], v
                yield {"points": pts, "point_scores": sc, "score": 0.0 if np.isnan(v) else v}
            return
        vals = self._host("pred_peak_values", b)
        if vals is None:
            vals = np.full(kp.shape[:2], np.nan, dtype=np.float32)
        iscore = self._host("instance_scores", b)
        tscore = self._host("instance_tracking_scores", b)
        cls = self._host("pred_class_inds", b)
        for i in range(kp.shape[0]):
            if np.isnan(kp[i]).all():
                continue  # NaN-padded slot
            # no instance score (single-instance models): the SUM of the node confidences, nansum of an all-NaN row = 0 (:388-397)
            row = {"points": kp[i], "point_scores": vals[i], "score": float(iscore[i]) if iscore is not None else float(np.nansum(vals[i]))}
            if n_tracks is not None:
                # top-down multi-class carries the class per instance (node 0 of pred_class_inds); bottom-up multi-class: slot i IS the class
                c = int(cls[i, 0]) if cls is not None else i
                if 0 <= c < n_tracks:
                    row["track_index"] = c
                if tscore is not None:
                    row["tracking_score"] = float(tscore[i])
            yield row

    def to_instances(self, skeleton, batch_index: int = 0, anchor_ind: Optional[int] = None, tracks: Optional[list] = None, *,
                     collapse_skeleton=None) -> list:
        """``sio.PredictedInstance`` per non-NaN instance slot of one sample (outputs.py:284-432).  ``skeleton``: a
        ``sleap_io.Skeleton``; ``tracks``: per-class ``sio.Track`` list of the multi-class models; ``collapse_skeleton``: a 1-node
        skeleton that a stand-alone centroid model's output is packaged on instead of NaN-padding ``skeleton``."""
        import sleap_io as sio

        if self.pred_keypoints is None and self.pred_centroids is not None and collapse_skeleton is not None:
            skeleton, anchor_ind = collapse_skeleton, 0
        out = []
        for row in self._instance_rows(batch_index, len(skeleton.nodes), anchor_ind, None if tracks is None else len(tracks)):
            kw = {}
            if "track_index" in row:
                kw["track"] = tracks[row["track_index"]]
            if "tracking_score" in row:
                kw["tracking_score"] = row["tracking_score"]
            out.append(sio.PredictedInstance.from_numpy(points_data=row["points"], point_scores=row["point_scores"], score=row["score"], skeleton=skeleton, **kw))
        return out

    def to_labels(self, skeleton, videos: Optional[list] = None, anchor_ind: Optional[int] = None, tracks: Optional[list] = None, *,
                  collapse_skeleton=None, keep_empty_frames: bool = False):
        """``sio.Labels`` with one ``LabeledFrame`` per batch slot that has instances (every slot with ``keep_empty_frames``),
        frame / video indices from ``frame_indices`` / ``video_indices`` (outputs.py:612-779; pose fields only: no masks / ROIs)."""
        import sleap_io as sio

        videos = list(videos) if videos else [None]
        frames: List[Any] = []
        used: List[Any] = []
        for b in range(self.batch_size):
            inst = self.to_instances(skeleton, b, anchor_ind, tracks, collapse_skeleton=collapse_skeleton)
            if not inst and not keep_empty_frames:
                continue
            for it in inst:
                t = getattr(it, "track", None)
                if t is not None and all(t is not u for u in used):
                    used.append(t)
            vi = int(self.video_indices[b]) if self.video_indices is not None else 0
            if vi >= len(videos) and len(videos) != 1:  # a provider / packaging mismatch must be loud, not wrap onto another video
                raise IndexError(f"video_index {vi} is out of range for {len(videos)} videos; the provider emitted a video index with no matching video.")
            frames.append(sio.LabeledFrame(video=videos[vi] if vi < len(videos) else videos[0], frame_idx=int(self.frame_indices[b]) if self.frame_indices is not None else b, instances=inst))
        pkg = collapse_skeleton if collapse_skeleton is not None else skeleton
        labels = sio.Labels(labeled_frames=frames, videos=[v for v in videos if v is not None], skeletons=[pkg] if pkg is not None else [])
        if used:
            labels.tracks = used
        return labels

    def __repr__(self) -> str:
        parts = []
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, torch.Tensor):
                parts.append(f"{f.name}=Tensor{tuple(v.shape)}")
        return f"Outputs({', '.join(parts)})" if parts else "Outputs(empty)"
