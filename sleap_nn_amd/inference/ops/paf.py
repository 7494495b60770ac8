"""PAF grouping: device line-integral scoring + C++ host matching/assembly.

API mirror of ``sleap_nn/inference/ops/paf.py`` (PAFScorer :1152-1532 and the free
functions it wraps).  The arithmetic lives in csrc/post_kernels.hip (scoring) and
csrc/group_host.cpp (assignment + assembly); this module only marshals tensors.

One documented difference: the reference enumerates the candidates of an edge through an
*unstable* ``torch.argsort`` (paf.py:108), so their order inside an edge depends on the
machine's sort kernel; here the order is always (edge, src peak, dst peak) ascending.
Matching and grouping do not depend on that order.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from sleap_nn_amd import _lib as L


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def toposort_edges(edge_inds: Sequence[Tuple[int, int]]) -> Tuple[int, ...]:
    """paf.py:890-912 (networkx topological root + BFS edge order), in C++."""
    e = _i32(np.asarray(list(edge_inds), dtype=np.int32).reshape(-1, 2))
    out = np.zeros(max(1, e.shape[0]), dtype=np.int32)
    n = L.check(L.lib().ph_toposort_edges(C.c_void_p(e.ctypes.data), e.shape[0], C.c_void_p(out.ctypes.data)))
    return tuple(int(v) for v in out[:n])


def linear_sum_assignment(cost) -> Tuple[np.ndarray, np.ndarray]:
    """Drop-in for ``scipy.optimize.linear_sum_assignment`` (minimise) as used at paf.py:589."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    if cost.ndim != 2:
        raise ValueError("expected a matrix (2-D array), got a %r array" % (cost.shape,))
    nr, nc = cost.shape
    n = min(nr, nc)
    rows = np.zeros(max(n, 1), dtype=np.int32)
    cols = np.zeros(max(n, 1), dtype=np.int32)
    rc = L.lib().ph_lsap(C.c_void_p(cost.ctypes.data), nr, nc, C.c_void_p(rows.ctypes.data), C.c_void_p(cols.ctypes.data))
    if rc < 0:
        raise ValueError(L.lib().ph_last_error().decode())
    return rows[:n].astype(np.int64), cols[:n].astype(np.int64)


def score_paf_lines_device(pafs_nchw: torch.Tensor, peaks_xy: torch.Tensor, peak_channel: torch.Tensor, peak_offsets: torch.Tensor,
                           n_peaks_total: int, edge_inds: torch.Tensor, n_nodes: int, n_points: int, pafs_stride: int,
                           max_edge_length_ratio: float, dist_penalty_weight: float, capacity: int, out=None):
    """Enqueue candidate enumeration + line scoring; everything stays on the device.

    Returns ``(cand_edge, cand_src, cand_dst, cand_score, cand_offsets int32[B+1])``; ``out``: the same five as
    pre-allocated views (a caller's packed D2H arena) the kernels write directly.
    """
    L.require_cuda(pafs_nchw, "pafs")
    pafs = pafs_nchw.detach().to(torch.float32).contiguous()
    B, E2, H, W = pafs.shape
    dev = pafs.device
    n_edges = E2 // 2
    # max() includes the channel axis, as in the reference (paf.py:457-461)
    max_edge_length = float(max_edge_length_ratio * max(E2, W, H) * pafs_stride)
    cap = int(capacity)
    if out is not None:
        ce, cs, cd, sc, off = out
        assert ce.numel() == cap and cs.numel() == cap and cd.numel() == cap and sc.numel() == cap and off.numel() == B + 1
    else:
        ce = torch.empty((cap,), dtype=torch.int32, device=dev)
        cs = torch.empty((cap,), dtype=torch.int32, device=dev)
        cd = torch.empty((cap,), dtype=torch.int32, device=dev)
        sc = torch.empty((cap,), dtype=torch.float32, device=dev)
        off = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    t = _linspace_on(dev, n_points)  # exact torch.linspace values (paf.py:190)
    scratch = torch.empty((n_peaks_total + B * (n_nodes + 1) + B * (n_edges + 1) + B + 16,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        L.check(
            L.lib().ph_paf_score(
                C.c_void_p(pafs.data_ptr()), B, E2, H, W, C.c_void_p(peaks_xy.data_ptr()), C.c_void_p(peak_channel.data_ptr()),
                C.c_void_p(peak_offsets.data_ptr()), int(n_peaks_total), int(n_nodes), C.c_void_p(edge_inds.data_ptr()), n_edges,
                C.c_void_p(t.data_ptr()), int(n_points), int(pafs_stride), max_edge_length, float(dist_penalty_weight),
                C.c_void_p(ce.data_ptr()), C.c_void_p(cs.data_ptr()), C.c_void_p(cd.data_ptr()), C.c_void_p(sc.data_ptr()),
                C.c_void_p(off.data_ptr()), cap, C.c_void_p(scratch.data_ptr()), scratch.numel() * 4, L.current_stream_ptr(),
            )
        )
    return ce, cs, cd, sc, off


_LINSPACE = {}


def _linspace_on(dev, n_points: int) -> torch.Tensor:
    """``torch.linspace(0, 1, n_points)`` computed once on the host (the reference's values) and kept on the device."""
    key = (str(dev), int(n_points))
    if key not in _LINSPACE:
        _LINSPACE[key] = torch.linspace(0, 1, steps=n_points).to(dev)
    return _LINSPACE[key]


def group_batch_host(n_nodes: int, edge_inds, peaks_xy: np.ndarray, peak_vals: np.ndarray, peak_channel: np.ndarray, peak_offsets: np.ndarray,
                     cand_edge: np.ndarray, cand_src: np.ndarray, cand_dst: np.ndarray, cand_score: np.ndarray, cand_offsets: np.ndarray,
                     min_line_scores: float, min_instance_peaks: Union[int, float], max_instances: int, truncate_by_score: bool):
    """One C++ call for a whole batch: per-edge assignment + greedy assembly + NaN padding."""
    B = len(peak_offsets) - 1
    e = _i32(np.asarray(list(edge_inds)).reshape(-1, 2))
    pk = np.ascontiguousarray(peaks_xy, dtype=np.float32)
    pv = np.ascontiguousarray(peak_vals, dtype=np.float32)
    pc, po = _i32(peak_channel), _i32(peak_offsets)
    ce, cs, cd, co = _i32(cand_edge), _i32(cand_src), _i32(cand_dst), _i32(cand_offsets)
    sc = np.ascontiguousarray(cand_score, dtype=np.float32)
    kp = np.empty((B, max_instances, n_nodes, 2), dtype=np.float32)
    vals = np.empty((B, max_instances, n_nodes), dtype=np.float32)
    scores = np.empty((B, max_instances), dtype=np.float32)
    n_inst = np.zeros((B,), dtype=np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = L.lib().ph_group_batch(
        B, n_nodes, p(e), e.shape[0], p(pk), p(pv), p(pc), p(po), p(ce), p(cs), p(cd), p(sc), p(co), float(min_line_scores),
        float(min_instance_peaks), 1 if isinstance(min_instance_peaks, float) else 0, int(max_instances), 1 if truncate_by_score else 0,
        p(kp), p(vals), p(scores), p(n_inst),
    )
    if rc == L.PH_E_INFEASIBLE:
        raise ValueError("cost matrix is infeasible")  # what scipy raises inside the reference
    L.check(rc)
    return kp, vals, scores, n_inst


class PAFScorer:
    """Parameter bundle + high-level grouping API (paf.py:1152-1532)."""

    def __init__(self, part_names: List[str], edges: List[Tuple[str, str]], pafs_stride: int, max_edge_length_ratio: float = 0.25,
                 dist_penalty_weight: float = 1.0, n_points: int = 10, min_instance_peaks: Union[int, float] = 0, min_line_scores: float = 0.25):
        self.part_names = list(part_names)
        self.edges = [tuple(e) for e in edges]
        self.pafs_stride = pafs_stride
        self.max_edge_length_ratio = max_edge_length_ratio
        self.dist_penalty_weight = dist_penalty_weight
        self.n_points = n_points
        self.min_instance_peaks = min_instance_peaks
        self.min_line_scores = min_line_scores
        self.edge_inds = [(self.part_names.index(s), self.part_names.index(d)) for s, d in self.edges]
        self.n_nodes = len(self.part_names)
        self.n_edges = len(self.edges)
        self.sorted_edge_inds = toposort_edges(self.edge_inds)
        self._edges_dev = {}

    @classmethod
    def from_config(cls, config, max_edge_length_ratio=0.25, dist_penalty_weight=1.0, n_points=10, min_instance_peaks=0, min_line_scores=0.25):
        from sleap_nn_amd.utils import cfg_get

        cm, pf = cfg_get(config, "confmaps"), cfg_get(config, "pafs")
        return cls(
            part_names=list(cfg_get(cm, "part_names")), edges=[tuple(e) for e in cfg_get(pf, "edges")], pafs_stride=cfg_get(pf, "output_stride"),
            max_edge_length_ratio=max_edge_length_ratio, dist_penalty_weight=dist_penalty_weight, n_points=n_points,
            min_instance_peaks=min_instance_peaks, min_line_scores=min_line_scores,
        )

    def edges_on(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._edges_dev:
            self._edges_dev[key] = torch.tensor(self.edge_inds, dtype=torch.int32).reshape(-1, 2).contiguous().to(device)
        return self._edges_dev[key]

    # -- reference-shaped API (lists of per-sample tensors) -----------------------------------
    def score_paf_lines(self, pafs: torch.Tensor, peaks: List[torch.Tensor], peak_channel_inds: List[torch.Tensor]):
        """``pafs``: (B, H, W, 2E) as in the reference.  Returns per-sample lists
        ``(edge_inds, edge_peak_inds (n,2), line_scores)`` on the device."""
        nchw = pafs.permute(0, 3, 1, 2)
        dev = nchw.device
        B = nchw.shape[0]
        lens = [int(p.shape[0]) for p in peaks]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        n_tot = int(offs[-1])
        xy = torch.cat([p.reshape(-1, 2).to(dev, torch.float32) for p in peaks]).contiguous() if n_tot else torch.zeros((1, 2), device=dev)
        ch = torch.cat([c.to(dev, torch.int32) for c in peak_channel_inds]).contiguous() if n_tot else torch.zeros((1,), dtype=torch.int32, device=dev)
        cap = 1
        for b in range(B):
            cnt = np.bincount(peak_channel_inds[b].detach().cpu().numpy().astype(np.int64), minlength=self.n_nodes) if lens[b] else np.zeros(self.n_nodes, dtype=np.int64)
            cap += int(sum(cnt[s] * cnt[d] for s, d in self.edge_inds))
        ce, cs, cd, sc, off = score_paf_lines_device(
            nchw, xy, ch, torch.from_numpy(offs).to(dev), n_tot, self.edges_on(dev), self.n_nodes, self.n_points, self.pafs_stride,
            self.max_edge_length_ratio, self.dist_penalty_weight, cap,
        )
        off_h = off.cpu().numpy()
        e_l, p_l, s_l = [], [], []
        for b in range(B):
            a, z = int(off_h[b]), int(off_h[b + 1])
            e_l.append(ce[a:z])
            p_l.append(torch.stack([cs[a:z], cd[a:z]], dim=1).to(torch.int64))
            s_l.append(sc[a:z])
        return e_l, p_l, s_l

    def match_candidates(self, edge_inds, edge_peak_inds, line_scores):
        """paf.py:500-702 with the native assignment solver (per-sample lists in, lists out, CPU)."""
        me, ms, md, msc = [], [], [], []
        for b in range(len(edge_inds)):
            e = edge_inds[b].detach().cpu().numpy()
            pr = edge_peak_inds[b].detach().cpu().numpy().reshape(-1, 2)
            sc = line_scores[b].detach().cpu().numpy().astype(np.float32)
            oe, os_, od, osc = [], [], [], []
            for k in range(self.n_edges):
                sel = np.nonzero(e == k)[0]
                if sel.size == 0:
                    continue
                su, du = np.unique(pr[sel, 0]), np.unique(pr[sel, 1])
                cost = np.full((su.size, du.size), np.inf, dtype=np.float32)
                cost[np.searchsorted(su, pr[sel, 0]), np.searchsorted(du, pr[sel, 1])] = -sc[sel]
                cost[np.isnan(cost)] = np.inf
                r, c = linear_sum_assignment(cost)
                oe.append(np.full(r.size, k, dtype=np.int32))
                os_.append(r.astype(np.int32))
                od.append(c.astype(np.int32))
                osc.append((-cost[r, c]).astype(np.float32))
            cat = lambda xs, dt: torch.from_numpy(np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt))
            me.append(cat(oe, np.int32))
            ms.append(cat(os_, np.int32))
            md.append(cat(od, np.int32))
            msc.append(cat(osc, np.float32))
        return me, ms, md, msc

    def predict(self, pafs: torch.Tensor, peaks: List[torch.Tensor], peak_vals: List[torch.Tensor], peak_channel_inds: List[torch.Tensor]):
        """paf.py:1469-1532: returns per-sample lists (instances (n,N,2), peak scores (n,N), instance scores (n,)),
        plus the scored-candidate lists, like the reference's 6-tuple."""
        e_l, p_l, s_l = self.score_paf_lines(pafs, peaks, peak_channel_inds)
        B = len(peaks)
        lens = [int(p.shape[0]) for p in peaks]
        po = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        co = np.concatenate([[0], np.cumsum([int(x.shape[0]) for x in e_l])]).astype(np.int32)
        cat = lambda xs, dt, w=None: (np.concatenate([x.detach().cpu().numpy().reshape(-1, *( [w] if w else [])) for x in xs]).astype(dt) if len(xs) and sum(x.numel() for x in xs) else np.zeros((0, w) if w else (0,), dtype=dt))
        pk, pv, pc = cat(peaks, np.float32, 2), cat(peak_vals, np.float32), cat(peak_channel_inds, np.int32)
        ce, sc = cat(e_l, np.int32), cat(s_l, np.float32)
        pr = cat(p_l, np.int32, 2)
        max_inst = max(1, max(lens) if lens else 1)
        kp, vals, scores, n_inst = group_batch_host(self.n_nodes, self.edge_inds, pk, pv, pc, po, ce, pr[:, 0] if pr.size else pr.reshape(-1), pr[:, 1] if pr.size else pr.reshape(-1), sc, co, self.min_line_scores, self.min_instance_peaks, max_inst, False)
        inst = [torch.from_numpy(kp[b, : n_inst[b]].copy()) for b in range(B)]
        ivals = [torch.from_numpy(vals[b, : n_inst[b]].copy()) for b in range(B)]
        iscores = [torch.from_numpy(scores[b, : n_inst[b]].copy()) for b in range(B)]
        return inst, ivals, iscores, e_l, p_l, s_l
