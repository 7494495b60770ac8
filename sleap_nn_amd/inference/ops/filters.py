"""Post-inference instance filters on ``Outputs`` arrays (host side, numpy).

Array form of the reference's ``sleap_nn/inference/ops/filters.py`` (which walks ``sio.Labels``):
the same decisions per frame -- ``filter_by_node_count`` :13-91, ``filter_by_node_confidence``
:94-170, ``filter_overlapping_instances`` :224-299 with greedy NMS on bounding-box IoU
(:336-374, :419-448) or on the simplified OKS (:377-416, :451-495) -- applied to the
``(B, I, N, 2)`` keypoints / ``(B, I, N)`` peak values / ``(B, I)`` scores of an ``Outputs``.
Removed instances become NaN rows (the shape convention of ``Outputs``); ``keep`` lists give the
surviving instance indices per frame in the reference's order (decreasing score for the NMS).
"""
from __future__ import annotations

from dataclasses import replace
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from sleap_nn_amd.inference.outputs import Outputs


def _present(kp: np.ndarray) -> np.ndarray:
    """(I,) instances that exist at all (at least one non-NaN node)."""
    return ~np.isnan(kp).all(axis=(1, 2))


def visible_nodes(kp: np.ndarray) -> np.ndarray:
    """filters.py:173-184: nodes whose (x, y) are both finite, counted per instance."""
    return (~np.isnan(kp).any(axis=-1)).sum(axis=-1)


def mean_node_score(kp: np.ndarray, vals: np.ndarray) -> np.ndarray:
    """filters.py:187-221: mean peak value over visible nodes, NaN scores skipped, 0.0 if none."""
    out = np.zeros(kp.shape[0], dtype=np.float64)
    for i in range(kp.shape[0]):
        valid = ~np.isnan(kp[i]).any(axis=1)
        s = vals[i][valid]
        s = s[~np.isnan(s)]
        out[i] = float(np.mean(s)) if len(s) else 0.0
    return out


def instance_bbox(pts: np.ndarray) -> np.ndarray:
    """filters.py:302-321: [xmin, ymin, xmax, ymax] over visible nodes, zeros if there are none."""
    valid = ~np.isnan(pts).any(axis=1)
    if not valid.any():
        return np.array([0.0, 0.0, 0.0, 0.0])
    p = pts[valid]
    return np.array([p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max()])


def iou_one_to_many(box: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """filters.py:419-448."""
    iw = np.maximum(0.0, np.minimum(box[2], boxes[:, 2]) - np.maximum(box[0], boxes[:, 0]))
    ih = np.maximum(0.0, np.minimum(box[3], boxes[:, 3]) - np.maximum(box[1], boxes[:, 1]))
    inter = iw * ih
    union = (box[2] - box[0]) * (box[3] - box[1]) + (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(union > 0, inter / union, 0.0)


def oks(a: np.ndarray, b: np.ndarray, kappa: float = 0.1) -> float:
    """filters.py:451-495: equal-weight OKS, scale = bbox area of ``a``'s visible nodes."""
    va, vb = ~np.isnan(a).any(axis=1), ~np.isnan(b).any(axis=1)
    valid = va & vb
    if not valid.any():
        return 0.0
    pa = a[va]
    if len(pa) < 2:
        return 0.0
    scale_sq = (pa[:, 0].max() - pa[:, 0].min()) * (pa[:, 1].max() - pa[:, 1].min())
    if scale_sq <= 0:
        return 0.0
    d_sq = np.sum((a[valid] - b[valid]) ** 2, axis=1)
    return float(np.mean(np.exp(-d_sq / (2 * scale_sq * kappa**2))))


def nms_greedy(points: Sequence[np.ndarray], scores: np.ndarray, threshold: float, method: str = "iou") -> List[int]:
    """filters.py:336-416: keep the best, drop everything more similar than ``threshold`` to it, repeat."""
    if len(points) == 0:
        return []
    if method not in ("iou", "oks"):
        raise ValueError(f"Unknown method: {method}. Use 'iou' or 'oks'.")
    order = np.asarray(scores).argsort()[::-1].tolist()
    boxes = np.array([instance_bbox(p) for p in points]) if method == "iou" else None
    keep: List[int] = []
    while order:
        i = order.pop(0)
        keep.append(i)
        if not order:
            break
        if method == "iou":
            sim = iou_one_to_many(boxes[i], boxes[np.array(order)])
        else:
            sim = np.array([oks(points[i], points[j]) for j in order])
        order = [order[j] for j in range(len(order)) if sim[j] <= threshold]
    return keep


def _apply(outputs: Outputs, keep: List[List[int]]) -> Outputs:
    kp = outputs.pred_keypoints.clone()
    B, I = kp.shape[:2]
    drop = torch.ones((B, I), dtype=torch.bool)
    for b, ks in enumerate(keep):
        drop[b, ks] = False
    kw = {"pred_keypoints": kp}
    kp[drop] = float("nan")
    for name in ("pred_peak_values", "instance_scores", "pred_crop_keypoints", "pred_centroids", "pred_centroid_values", "instance_tracking_scores"):
        t = getattr(outputs, name)
        if isinstance(t, torch.Tensor) and t.shape[:2] == (B, I):
            t = t.clone()
            t[drop] = float("nan")
            kw[name] = t
    return replace(outputs, **kw)


def filter_outputs(outputs: Outputs, min_visible_nodes: int = 0, min_visible_node_fraction: float = 0.0, min_instance_score: float = 0.0,
                   min_mean_node_score: float = 0.0, overlap_threshold: Optional[float] = None, overlap_method: str = "iou") -> Tuple[Outputs, List[List[int]]]:
    """Run the three filters in the reference's order (node count, confidence, overlap) on every frame."""
    o = outputs.cpu()
    kp = o.pred_keypoints.numpy().astype(np.float64)
    vals = o.pred_peak_values.numpy().astype(np.float64) if o.pred_peak_values is not None else None
    scores = o.instance_scores.numpy().astype(np.float64) if o.instance_scores is not None else None
    n_nodes = kp.shape[2]
    keep_all: List[List[int]] = []
    for b in range(kp.shape[0]):
        idx = [i for i in np.nonzero(_present(kp[b]))[0].tolist()]
        if min_visible_nodes > 0 or min_visible_node_fraction > 0.0:
            nv = visible_nodes(kp[b])
            idx = [i for i in idx if not (min_visible_nodes > 0 and nv[i] < min_visible_nodes)
                   and not (min_visible_node_fraction > 0.0 and (nv[i] / n_nodes if n_nodes else 0.0) < min_visible_node_fraction)]
        if min_instance_score > 0.0 or min_mean_node_score > 0.0:
            ms = mean_node_score(kp[b], vals[b]) if (vals is not None and min_mean_node_score > 0.0) else None
            sc = scores[b] if scores is not None else np.ones(kp.shape[1])
            idx = [i for i in idx if not (min_instance_score > 0.0 and (1.0 if np.isnan(sc[i]) and scores is None else sc[i]) < min_instance_score)
                   and not (ms is not None and ms[i] < min_mean_node_score)]
        if overlap_threshold is not None and len(idx) > 1:
            sc = scores[b][idx] if scores is not None else np.ones(len(idx))
            k = nms_greedy([kp[b, i] for i in idx], sc, overlap_threshold, overlap_method)
            idx = [idx[j] for j in k]
        keep_all.append(idx)
    return _apply(o, keep_all), keep_all
