"""``BottomUpLayer``: confmaps + PAFs -> peaks -> scored candidates (GPU) -> grouping (CPU).

Mirror of ``sleap_nn/inference/layers/bottomup.py:44-236``.  The GPU stage enqueues three
native calls (local peaks, candidate scoring) with NO intermediate host sync, copies one
packed result arena to pinned host memory, synchronises once, and hands a flattened
``ScoredBatch`` to the C++ grouping stage.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from sleap_nn_amd.inference.backends import ModelBackend
from sleap_nn_amd.inference.layers.base import InferenceLayer
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig
from sleap_nn_amd.inference.ops.paf import PAFScorer, score_paf_lines_device
from sleap_nn_amd.inference.ops.peaks import find_local_peaks_device
from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo
from sleap_nn_amd.inference.streaming import GroupingParams, ScoredBatch, group_scored_batch


class BottomUpLayer(InferenceLayer):
    def __init__(self, backend: ModelBackend, paf_scorer: PAFScorer, cms_output_stride: int, pafs_output_stride: int,
                 max_instances: Optional[int] = None, max_stride: int = 1, max_peaks_per_node: Optional[int] = None,
                 preprocess_config: Optional[PreprocessConfig] = None, postprocess_config: Optional[PostprocessConfig] = None) -> None:
        super().__init__(backend, preprocess_config or PreprocessConfig(), postprocess_config or PostprocessConfig(), cms_output_stride, max_stride)
        self.paf_scorer = paf_scorer
        self.cms_output_stride = cms_output_stride
        self.pafs_output_stride = pafs_output_stride
        self.max_instances = max_instances
        self.max_peaks_per_node = max_peaks_per_node
        self._peak_cap = 0
        self._cand_cap = 0

    # -- GPU stage ------------------------------------------------------------------------
    def _enqueue_scoring(self, raw_out: dict, info: PreprocInfo) -> dict:
        """Enqueue peak finding + candidate scoring and ONE asynchronous D2H of the counts and of the
        (capacity-sized) payload into pinned memory; no host sync.  ``_finish_scoring`` turns the
        returned handle into a ``ScoredBatch`` once its event has fired, so a caller can keep the GPU
        busy with the next batch while this one drains (HIP stream order does the rest)."""
        cms = raw_out["MultiInstanceConfmapsHead"]
        pafs = raw_out["PartAffinityFieldsHead"]  # (B, 2E, H, W); the permute is folded into the kernel
        pc = self.postprocess_config
        sc = self.paf_scorer
        B, n_nodes = cms.shape[0], cms.shape[1]
        dev = cms.device
        peak_cap = max(self._peak_cap, B * n_nodes * 32, 1024)
        cand_cap = max(self._cand_cap, B * sc.n_edges * 256, 4096)
        # ONE packed arena, written by the kernels themselves and copied to the host in one piece (int32 rows travel bit-cast
        # as float32): [counts 2+2B | cand offsets B+1 | xy 2P | vals P | score Q | channel P | cand edge Q | src Q | dst Q]
        n_head = (2 + 2 * B) + (B + 1)
        packed = torch.empty(n_head + 4 * peak_cap + 4 * cand_cap, dtype=torch.float32, device=dev)
        ints = packed.view(torch.int32)
        o = n_head
        xy = packed[o : o + 2 * peak_cap].view(peak_cap, 2)
        o += 2 * peak_cap
        vals = packed[o : o + peak_cap]
        o += peak_cap
        score = packed[o : o + cand_cap]
        o += cand_cap
        ch = ints[o : o + peak_cap]
        ce, cs, cd = (ints[o + peak_cap + k * cand_cap : o + peak_cap + (k + 1) * cand_cap] for k in range(3))
        counts, coff = ints[: 2 + 2 * B], ints[2 + 2 * B : n_head]
        # peaks * cms_output_stride (bottomup.py:111) is the kernel's last multiply
        find_local_peaks_device(cms, pc.peak_threshold, pc.effective_refinement, pc.integral_patch_size, peak_cap,
                                xy_scale=float(self.cms_output_stride), out=(xy, vals, ch, counts))
        offs = counts[1 + B : 2 + 2 * B]
        score_paf_lines_device(
            pafs, xy, ch, offs, peak_cap, sc.edges_on(dev), sc.n_nodes, sc.n_points, sc.pafs_stride, sc.max_edge_length_ratio,
            sc.dist_penalty_weight, cand_cap, out=(ce, cs, cd, score, coff),
        )
        key = (int(packed.numel()), dev)
        pool = self.__dict__.setdefault("_pinned", {})
        host = pool.get(key, [])
        buf = host.pop() if host else torch.empty(packed.numel(), dtype=torch.float32, pin_memory=True)
        buf.copy_(packed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return {"buf": buf, "key": key, "event": ev, "B": B, "n_nodes": n_nodes, "peak_cap": int(peak_cap), "cand_cap": int(cand_cap), "n_head": int(n_head),
                "raw": raw_out, "info": info, "keep": packed}

    def _finish_scoring(self, h: dict) -> ScoredBatch:
        h["event"].synchronize()
        B, n_nodes, pcap, ccap = h["B"], h["n_nodes"], h["peak_cap"], h["cand_cap"]
        arr = h["buf"].numpy()
        head = arr[: h["n_head"]].view(np.int32)
        n_peaks = int(head[0])
        n_cand = int(head[2 + 2 * B + B])
        if n_peaks > pcap or n_cand > ccap:  # rare: capacity too small -> grow and redo this batch synchronously
            self._peak_cap = max(self._peak_cap, int(n_peaks * 1.25) + 16)
            self._cand_cap = max(self._cand_cap, int(n_cand * 1.25) + 16)
            self._pinned[h["key"]] = []
            return self._finish_scoring(self._enqueue_scoring(h["raw"], h["info"]))
        self._peak_cap, self._cand_cap = max(self._peak_cap, pcap), max(self._cand_cap, ccap)
        peak_offsets = head[1 + B : 2 + 2 * B].astype(np.int32)
        cand_offsets = head[2 + 2 * B :].astype(np.int32)
        o = h["n_head"]
        peaks_xy = arr[o : o + 2 * n_peaks].reshape(-1, 2).copy()
        o += 2 * pcap
        peak_vals = arr[o : o + n_peaks].copy()
        o += pcap
        cand_score = arr[o : o + n_cand].copy()
        o += ccap
        il = arr[o:].view(np.int32)
        peak_channel = il[:n_peaks].copy()
        cand_edge = il[pcap : pcap + n_cand].copy()
        cand_src = il[pcap + ccap : pcap + ccap + n_cand].copy()
        cand_dst = il[pcap + 2 * ccap : pcap + 2 * ccap + n_cand].copy()
        self._pinned.setdefault(h["key"], []).append(h["buf"])  # every view was copied out: the pinned buffer can be reused

        pc = self.postprocess_config
        cms, pafs = h["raw"]["MultiInstanceConfmapsHead"], h["raw"]["PartAffinityFieldsHead"]
        skip = False
        if self.max_peaks_per_node is not None:
            for b in range(B):
                c = peak_channel[peak_offsets[b] : peak_offsets[b + 1]]
                if c.size and int(np.bincount(c, minlength=n_nodes).max()) > self.max_peaks_per_node:
                    skip = True  # combinatorial-blow-up guard (bottomup.py:126-161)
                    break
        keep_cms = pc.return_confmaps or (pc.return_paf_graph and not skip)
        keep_pafs = pc.return_pafs or (pc.return_paf_graph and not skip)
        return ScoredBatch(
            peaks_xy=peaks_xy, peak_vals=peak_vals, peak_channel=peak_channel, peak_offsets=peak_offsets,
            cand_edge=cand_edge, cand_src=cand_src, cand_dst=cand_dst, cand_score=cand_score, cand_offsets=cand_offsets,
            info=h["info"].cpu(), n_samples=B, n_nodes=n_nodes, skip_paf=skip,
            cms=cms.detach().cpu() if keep_cms else None, pafs=pafs.detach().cpu() if keep_pafs else None,
        )

    def _score_pafs_on_gpu(self, raw_out: dict, info: PreprocInfo) -> ScoredBatch:
        """layers/bottomup.py:95-195: GPU stage with one sync (enqueue + finish back to back)."""
        return self._finish_scoring(self._enqueue_scoring(raw_out, info))

    def grouping_params(self) -> GroupingParams:
        max_instances = getattr(self.postprocess_config, "max_instances", None)
        if max_instances is None:
            max_instances = self.max_instances
        s = self.paf_scorer
        return GroupingParams(
            paf_scorer_kwargs={
                "part_names": list(s.part_names), "edges": [tuple(e) for e in s.edges], "pafs_stride": s.pafs_stride,
                "max_edge_length_ratio": s.max_edge_length_ratio, "dist_penalty_weight": s.dist_penalty_weight, "n_points": s.n_points,
                "min_instance_peaks": s.min_instance_peaks, "min_line_scores": s.min_line_scores,
            },
            max_instances=max_instances,
            return_confmaps=self.postprocess_config.return_confmaps,
            return_pafs=self.postprocess_config.return_pafs,
            return_paf_graph=self.postprocess_config.return_paf_graph,
        )

    def postprocess(self, raw_out: dict, info: PreprocInfo) -> Outputs:
        scored = self._score_pafs_on_gpu(raw_out, info)
        return group_scored_batch(scored, self.grouping_params())
