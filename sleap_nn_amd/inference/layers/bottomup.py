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
    def _score_pafs_on_gpu(self, raw_out: dict, info: PreprocInfo) -> ScoredBatch:
        cms = raw_out["MultiInstanceConfmapsHead"]
        pafs = raw_out["PartAffinityFieldsHead"]  # (B, 2E, H, W); the permute is folded into the kernel
        pc = self.postprocess_config
        sc = self.paf_scorer
        B, n_nodes = cms.shape[0], cms.shape[1]
        dev = cms.device
        peak_cap = max(self._peak_cap, B * n_nodes * 32, 1024)
        cand_cap = max(self._cand_cap, B * sc.n_edges * 256, 4096)
        while True:
            xy, vals, sb, ch, counts, _ = find_local_peaks_device(cms, pc.peak_threshold, pc.effective_refinement, pc.integral_patch_size, peak_cap)
            xy = xy * self.cms_output_stride  # peaks * cms_output_stride (bottomup.py:111)
            offs = counts[1 + B : 2 + 2 * B]
            ce, cs, cd, score, coff = score_paf_lines_device(
                pafs, xy, ch, offs, peak_cap, sc.edges_on(dev), sc.n_nodes, sc.n_points, sc.pafs_stride, sc.max_edge_length_ratio,
                sc.dist_penalty_weight, cand_cap,
            )
            # one packed D2H + one sync
            head = torch.cat([counts, coff]).cpu().numpy()
            n_peaks = int(head[0])
            n_cand = int(head[2 + 2 * B + B])
            if n_peaks > peak_cap or n_cand > cand_cap:
                peak_cap = max(peak_cap, int(n_peaks * 1.25) + 16)
                cand_cap = max(cand_cap, int(n_cand * 1.25) + 16)
                if n_peaks > xy.shape[0]:
                    continue  # peaks were truncated: candidates are incomplete, redo with room
                continue
            break
        self._peak_cap, self._cand_cap = peak_cap, cand_cap
        peak_offsets = head[1 + B : 2 + 2 * B].astype(np.int32)
        cand_offsets = head[2 + 2 * B :].astype(np.int32)
        # one packed D2H for all payload arrays (int32 rows travel bit-cast as float32)
        packed = torch.cat([xy[:n_peaks].reshape(-1), vals[:n_peaks], score[:n_cand],
                            torch.cat([ch[:n_peaks], ce[:n_cand], cs[:n_cand], cd[:n_cand]]).view(torch.float32)]).cpu().numpy()
        n_fl = 3 * n_peaks + n_cand
        fl = packed[:n_fl]
        il = packed[n_fl:].view(np.int32)
        peaks_xy = fl[: 2 * n_peaks].reshape(-1, 2)
        peak_vals = fl[2 * n_peaks : 3 * n_peaks]
        cand_score = fl[3 * n_peaks :]
        peak_channel = il[:n_peaks]
        cand_edge, cand_src, cand_dst = il[n_peaks : n_peaks + n_cand], il[n_peaks + n_cand : n_peaks + 2 * n_cand], il[n_peaks + 2 * n_cand :]

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
            info=info.cpu(), n_samples=B, n_nodes=n_nodes, skip_paf=skip,
            cms=cms.detach().cpu() if keep_cms else None, pafs=pafs.detach().cpu() if keep_pafs else None,
        )

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
