"""Pose-estimation metrics on plain arrays and the epoch-end evaluation hook of the training loop.

Restates the arithmetic of ``sleap_nn/evaluation.py`` (``compute_instance_area`` :626-641, ``compute_oks`` :644-760,
``match_instances`` :763-856, ``compute_dists`` :904-939 and the ``Evaluator`` metrics ``voc_metrics`` :1253-1362,
``mOKS`` :1364-1367, ``distance_metrics`` :1369-1400, ``pck_metrics`` :1824-1862) without sleap-io objects: a frame is a
``(n_instances, n_nodes, 2)`` array with NaN for missing points, a prediction additionally carries one score per instance.
``EpochEndEvaluator`` is the counterpart of the per-batch collection in the LightningModules' ``validation_step``
(``training/lightning_modules.py:1099-1142``) and of ``training/callbacks.py:1263-1323``: predictions arrive in original image
space, ground truth is divided by the sample's ``eff_scale``; metrics are computed once per epoch on the host.
Host-side NumPy by design: a few hundred instances per epoch, nothing for a GPU to do.
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def compute_instance_area(points: np.ndarray) -> np.ndarray:
    """Bounding-box area of each instance's visible points (evaluation.py:626-641)."""
    pts = np.asarray(points)
    if pts.ndim == 2:
        pts = pts[None]
    return np.prod(np.nanmax(pts, axis=-2) - np.nanmin(pts, axis=-2), axis=-1)


def compute_oks(points_gt: np.ndarray, points_pr: np.ndarray, scale=None, stddev=0.025, use_cocoeval: bool = True) -> np.ndarray:
    """Object keypoint similarity of every (ground truth, prediction) pair -> ``(n_gt, n_pr)`` (evaluation.py:644-760)."""
    gt = np.asarray(points_gt, dtype=np.float64 if np.asarray(points_gt).dtype == np.float64 else np.asarray(points_gt).dtype)
    pr = np.asarray(points_pr)
    if gt.ndim == 2:
        gt = gt[None]
    if pr.ndim == 2:
        pr = pr[None]
    n_gt, n_nodes, _ = gt.shape
    area = compute_instance_area(gt) if scale is None else (np.full(n_gt, scale) if np.isscalar(scale) else np.asarray(scale))
    sd = np.full(n_nodes, stddev) if np.isscalar(stddev) else np.asarray(stddev)
    d2 = ((gt[:, None] - pr[None]) ** 2).sum(axis=-1)  # (n_gt, n_pr, n_nodes) squared distances
    if use_cocoeval:
        norm = ((2 * sd) ** 2)[None, None, :] * (2 * (area + np.spacing(1)))[:, None, None]
    else:
        norm = (sd**2)[None, None, :] * (2 * ((area + np.spacing(1)) ** 2))[:, None, None]
    d2[:, np.isnan(pr).any(axis=-1)] = np.inf  # a missing predicted point is a miss
    ks = np.exp(-(d2 / norm))
    missing_gt = np.isnan(gt).any(axis=-1)
    ks[np.broadcast_to(missing_gt[:, None, :], ks.shape)] = 0  # invisible ground truth does not count
    n_visible = (~missing_gt).astype("float32").sum(axis=-1, keepdims=True)
    return ks.sum(axis=-1) / n_visible


def match_instances(gt: np.ndarray, pr: np.ndarray, pr_scores: np.ndarray, stddev=0.025, scale=None, threshold: float = 0.0):
    """PASCAL-VOC style greedy matching inside one frame (evaluation.py:763-856): predictions in descending score order
    (stable), each takes the available ground-truth instance with the best OKS above ``threshold``.

    Returns ``(pairs, false_negatives)``: ``pairs`` = list of ``(gt index, prediction index, oks)``, ``false_negatives`` = the
    unmatched ground-truth indices."""
    gt = np.asarray(gt)
    pr = np.asarray(pr)
    available = list(range(len(gt)))
    pairs: List[Tuple[int, int, float]] = []
    if len(gt) == 0:
        return pairs, available
    for ip in np.argsort(-np.asarray(pr_scores, dtype=np.float64), kind="mergesort"):
        oks = compute_oks(gt[available], pr[ip : ip + 1], stddev=stddev, scale=scale)[:, 0]
        oks[oks <= threshold] = np.nan
        best = int(np.argsort(-oks, kind="mergesort")[0])
        if np.isnan(oks[best]):
            continue
        pairs.append((available.pop(best), int(ip), float(oks[best])))
        if not available:
            break
    return pairs, available


class Evaluator:
    """Metrics over matched frames (evaluation.py:942-1016 restricted to the pose metrics of ``evaluate`` :1893-1941)."""

    def __init__(self, frames_gt: Sequence[np.ndarray], frames_pr: Sequence[np.ndarray], frames_pr_scores: Sequence[np.ndarray], oks_stddev: float = 0.025,
                 oks_scale: Optional[float] = None, match_threshold: float = 0.0) -> None:
        self.pair_oks: List[float] = []
        self.pair_scores: List[float] = []
        dists = []
        self.n_false_negatives = 0
        for g, p, s in zip(frames_gt, frames_pr, frames_pr_scores):
            g, p, s = np.asarray(g), np.asarray(p), np.asarray(s)
            keep_g = ~np.isnan(g).all(axis=(1, 2)) if len(g) else np.zeros(0, bool)
            keep_p = ~np.isnan(p).all(axis=(1, 2)) if len(p) else np.zeros(0, bool)
            g, p, s = g[keep_g], p[keep_p], s[keep_p]
            if len(g) == 0 or len(p) == 0:  # find_frame_pairs only pairs frames present on both sides (evaluation.py:558-623)
                continue
            pairs, fn = match_instances(g, p, s, stddev=oks_stddev, scale=oks_scale, threshold=match_threshold)
            self.n_false_negatives += len(fn)
            for ig, ip, oks in pairs:
                self.pair_oks.append(oks)
                self.pair_scores.append(float(s[ip]))
                dists.append(np.linalg.norm(p[ip] - g[ig], axis=-1))
        self.dists = np.array(dists)

    def mOKS(self) -> Dict[str, float]:
        o = np.array(self.pair_oks)
        return {"mOKS": float(o.mean()) if o.size else float("nan")}

    def voc_metrics(self, match_score_thresholds=np.linspace(0.5, 0.95, 10), recall_thresholds=np.linspace(0, 1, 101)) -> Dict[str, object]:
        name = "oks_voc"
        order = np.argsort(-np.array(self.pair_scores), kind="mergesort")
        match_scores = np.array(self.pair_oks)[order]
        npig = len(self.pair_oks) + self.n_false_negatives
        if match_scores.size == 0:
            return {f"{name}.{k}": 0 for k in ("match_score_thresholds", "recall_thresholds", "match_scores", "precisions", "recalls", "AP", "AR", "mAP", "mAR")}
        precisions, recalls = [], []
        for thr in match_score_thresholds:
            tp = np.cumsum(match_scores >= thr)
            fp = np.cumsum(match_scores < thr)
            rc = tp / npig
            pr = tp / (fp + tp + np.spacing(1))
            for i in range(len(pr) - 1, 0, -1):  # monotone non-increasing precision envelope
                if pr[i] > pr[i - 1]:
                    pr[i - 1] = pr[i]
            inds = np.searchsorted(rc, recall_thresholds, side="left")
            prec = np.zeros(inds.shape)
            ok = inds < len(pr)
            prec[ok] = pr[inds[ok]]
            precisions.append(prec)
            recalls.append(rc[-1])
        precisions, recalls = np.array(precisions), np.array(recalls)
        return {f"{name}.match_score_thresholds": match_score_thresholds, f"{name}.recall_thresholds": recall_thresholds, f"{name}.match_scores": match_scores,
                f"{name}.precisions": precisions, f"{name}.recalls": recalls, f"{name}.AP": precisions.mean(axis=1), f"{name}.AR": recalls,
                f"{name}.mAP": precisions.mean(), f"{name}.mAR": recalls.mean()}

    def distance_metrics(self) -> Dict[str, object]:
        d = self.dists
        out = {"dists": d, "avg": float(np.nanmean(d)) if d.size and not np.all(np.isnan(d)) else float("nan")}
        ok = ~np.isnan(d) if d.size else np.zeros(0, bool)
        for p in (50, 75, 90, 95, 99):
            out[f"p{p}"] = float(np.percentile(d[ok], p)) if ok.any() else float("nan")
        return out

    def pck_metrics(self, thresholds=np.linspace(1, 10, 10)) -> Dict[str, object]:
        d = np.copy(self.dists)
        if d.size == 0:
            return {"thresholds": thresholds, "pcks": np.zeros((0, 0, len(thresholds)), bool), "mPCK_parts": np.array([]), "mPCK": float("nan"), "PCK@5": float("nan"), "PCK@10": float("nan")}
        d[np.isnan(d)] = np.inf
        pcks = d[..., None] < np.reshape(thresholds, (1, 1, -1))
        parts = pcks.mean(axis=0).mean(axis=-1)
        return {"thresholds": thresholds, "pcks": pcks, "mPCK_parts": parts, "mPCK": float(parts.mean()),
                "PCK@5": float(pcks[:, :, int(np.argmin(np.abs(thresholds - 5)))].mean()), "PCK@10": float(pcks[:, :, int(np.argmin(np.abs(thresholds - 10)))].mean())}

    def evaluate(self) -> Dict[str, object]:
        return {"voc_metrics": self.voc_metrics(), "mOKS": self.mOKS(), "distance_metrics": self.distance_metrics(), "pck_metrics": self.pck_metrics()}


class EpochEndEvaluator:
    """Collect (prediction, ground truth) per validation sample, evaluate at the end of the epoch.

    ``add_batch`` takes an ``Outputs`` of an inference layer (keypoints already in original image space) and the batch's
    ground-truth instances in PREPROCESSED space with their ``eff_scale`` and ``num_instances``
    (lightning_modules.py:1112-1141); ``compute`` returns the metrics dictionary and clears the lists
    (callbacks.py:1263-1323; only every ``eval_frequency``-th epoch evaluates)."""

    def __init__(self, oks_stddev: float = 0.025, oks_scale: Optional[float] = None, eval_frequency: int = 1) -> None:
        self.oks_stddev, self.oks_scale, self.eval_frequency = oks_stddev, oks_scale, int(eval_frequency)
        self._pred: List[np.ndarray] = []
        self._score: List[np.ndarray] = []
        self._gt: List[np.ndarray] = []

    def add_batch(self, outputs, gt_instances, eff_scale, num_instances) -> None:
        kp = np.asarray(outputs.pred_keypoints)
        vals = np.asarray(outputs.pred_peak_values)
        gt = np.asarray(gt_instances, dtype=np.float32)
        eff = np.asarray(eff_scale, dtype=np.float32).reshape(-1)
        for i in range(kp.shape[0]):
            k, v = kp[i], vals[i]
            if k.ndim == 2:  # single instance: (n_nodes, 2)
                k, v = k[None], v[None]
            g = gt[i]
            if g.ndim == 4:
                g = g[0]  # the n_samples axis
            self._pred.append(k)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)  # an all-NaN (padding) instance: nanmean -> NaN, dropped by the Evaluator
                self._score.append(np.nanmean(v, axis=-1))  # instance score = mean peak value (callbacks.py:1366-1370)
            self._gt.append((g / eff[i])[: int(num_instances[i])])

    def compute(self, epoch: int = 0) -> Optional[Dict[str, object]]:
        out = None
        if (epoch + 1) % self.eval_frequency == 0 and self._pred:
            out = Evaluator(self._gt, self._pred, self._score, self.oks_stddev, self.oks_scale).evaluate()
        self._pred, self._score, self._gt = [], [], []
        return out
