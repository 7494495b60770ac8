"""``BottomUpLayer``: confmaps + PAFs -> peaks -> scored candidates (GPU) -> grouping (CPU).

Mirror of ``sleap_nn/inference/layers/bottomup.py:44-236``.  The GPU stage enqueues three
native calls (local peaks, candidate scoring) with NO intermediate host sync, copies one
packed result arena to pinned host memory, synchronises once, and hands a flattened
``ScoredBatch`` to the C++ grouping stage.
"""
from __future__ import annotations

import threading
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
        self._gpu_lock = threading.RLock()

    # -- GPU stage ------------------------------------------------------------------------
    def _capacities(self, B: int, n_nodes: int):
        return max(self._peak_cap, B * n_nodes * 32, 1024), max(self._cand_cap, B * self.paf_scorer.n_edges * 256, 4096)

    def _scoring_launches(self, raw_out: dict, peak_cap: int, cand_cap: int) -> torch.Tensor:
        """Peak finding + candidate scoring of one batch into ONE packed arena, written by the kernels themselves and copied to the host in one piece (int32 rows travel
        bit-cast as float32): [counts 2+2B | cand offsets B+1 | xy 2P | vals P | score Q | channel P | cand edge Q | src Q | dst Q].  Launches only: no host sync,
        capturable in a graph."""
        cms = raw_out["MultiInstanceConfmapsHead"]
        pafs = raw_out["PartAffinityFieldsHead"]  # (B, 2E, H, W); the permute is folded into the kernel
        pc = self.postprocess_config
        sc = self.paf_scorer
        B = cms.shape[0]
        dev = cms.device
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
        return packed

    def _to_host_async(self, packed: torch.Tensor):
        """The arena into a pinned buffer of this layer's pool, asynchronously; the event fires when it has landed."""
        dev = packed.device
        key = (int(packed.numel()), dev)
        pool = self.__dict__.setdefault("_pinned", {})
        host = pool.get(key, [])
        buf = host.pop() if host else torch.empty(packed.numel(), dtype=torch.float32, pin_memory=True)
        buf.copy_(packed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return buf, key, ev

    def _enqueue_scoring(self, raw_out: dict, info: PreprocInfo) -> dict:
        """Enqueue peak finding + candidate scoring and ONE asynchronous D2H of the counts and of the
        (capacity-sized) payload into pinned memory; no host sync.  ``_finish_scoring`` turns the
        returned handle into a ``ScoredBatch`` once its event has fired, so a caller can keep the GPU
        busy with the next batch while this one drains (HIP stream order does the rest)."""
        cms = raw_out["MultiInstanceConfmapsHead"]
        B, n_nodes = cms.shape[0], cms.shape[1]
        peak_cap, cand_cap = self._capacities(B, n_nodes)
        packed = self._scoring_launches(raw_out, peak_cap, cand_cap)
        buf, key, ev = self._to_host_async(packed)
        return {"buf": buf, "key": key, "event": ev, "B": B, "n_nodes": n_nodes, "peak_cap": int(peak_cap), "cand_cap": int(cand_cap), "n_head": int((2 + 2 * B) + (B + 1)),
                "raw": raw_out, "info": info, "keep": packed, "stream": torch.cuda.current_stream(cms.device)}

    def _enqueue_scoring_graphed(self, x: torch.Tensor, info: Optional[PreprocInfo] = None) -> dict:
        """The whole GPU stage of a batch -- forward, peak finding, candidate scoring -- as ONE hipGraph replay per preprocessed input shape (``InferenceLayer._graph_entry``),
        then the same asynchronous D2H as ``_enqueue_scoring``.  ``x``: preprocessed frames on the device with their ``info`` -- or, with ``info=None``, the uint8 batch BEFORE
        preprocessing: its resize / pad launches are captured in front of the forward (they depend on the batch's shape only).  Per batch the host issues a copy into the graph's input buffer, one
        graph launch, one D2H and one event -- at batch 4 of a small network the ~25 kernel launches of the eager stage cost more host time than the GPU needs to run them.
        The handle finishes through ``_finish_packed`` (or ``_finish_scoring``); a batch whose peaks overflow the captured capacities is redone eagerly with larger ones, and the
        next capture takes those."""
        be = self.backend
        pre = None
        if info is None:
            if x.dtype != torch.uint8:  # (float frames take normalize_on_gpu's data-dependent branch: a host read per batch; they are preprocessed outside the graph)
                x, info = self.preprocess(x)
            else:
                pre = self.preprocess
        x, code = be.input_code(x)
        B = int(x.shape[0])
        n_nodes = self.paf_scorer.n_nodes
        with self._gpu_lock:  # (a worker redoing an overflowed batch issues GPU work too: never while this thread captures or replays)
            peak_cap, cand_cap = self._capacities(B, n_nodes)

            def body(raw, _info):
                return self._scoring_launches(raw, peak_cap, cand_cap)

            graph, static_in, packed, _ws, info = self._graph_entry(x, info, code, body=body, extra_key=("gpu stage", int(peak_cap), int(cand_cap), self.cms_output_stride), pre=pre)
            if x.data_ptr() != static_in.data_ptr():
                static_in.copy_(x, non_blocking=True)
            graph.replay()
            buf, key, ev = self._to_host_async(packed)
        return {"buf": buf, "key": key, "event": ev, "B": B, "n_nodes": n_nodes, "peak_cap": int(peak_cap), "cand_cap": int(cand_cap), "n_head": int((2 + 2 * B) + (B + 1)),
                "raw": None, "x": x, "pre": pre is not None, "code": code, "info": info, "keep": packed, "stream": torch.cuda.current_stream(x.device)}

    def _redo_eagerly(self, h: dict) -> dict:
        """A handle whose capacities were exceeded (or whose maps are wanted back): the GPU stage again, kernel by kernel, with capacities grown from the counts it reported.
        May run on the host-stage worker thread: serialised against the enqueueing thread's captures / replays by ``_gpu_lock``, and complete (synchronised) on return."""
        with self._gpu_lock, torch.cuda.stream(h["stream"]):  # (on the stream the batch was enqueued on: the handle's workspace is in use there)
            raw = h["raw"]
            if raw is None:  # a graphed handle holds no head tensors (the graph's static ones have been overwritten since): run the forward again
                xin = self.preprocess(h["x"])[0] if h.get("pre") else h["x"]
                raw = self.backend.model.forward(xin.squeeze(1) if xin.dim() == 5 else xin, in_dtype=h["code"])
            self._pinned[h["key"]] = []
            h2 = self._enqueue_scoring(raw, h["info"])
            h2["event"].synchronize()
        return h2

    def _finish_packed(self, h: dict) -> Outputs:
        """Handle -> ``Outputs`` through ONE native call (``ph_group_packed``: unpack, capacity check, max_peaks_per_node guard, matching + assembly, scale undo, NaN pad) --
        what ``group_scored_batch(_finish_scoring(h), grouping_params())`` returns, without the per-field numpy copies and Python branches between them.  Falls back to exactly that
        pair when the caller wants the maps or the PAF graph back."""
        import ctypes as C

        from sleap_nn_amd import _lib as L

        pc = self.postprocess_config
        if pc.return_confmaps or pc.return_pafs or pc.return_paf_graph:
            if h["raw"] is None:
                h = self._redo_eagerly(h)
            return group_scored_batch(self._finish_scoring(h), self.grouping_params())
        h["event"].synchronize()
        B, n_nodes = h["B"], h["n_nodes"]
        sc = self.paf_scorer
        max_instances = getattr(pc, "max_instances", None)
        if max_instances is None:
            max_instances = self.max_instances
        if not max_instances:  # 0 = no cap, as `params.max_instances or _infer_max_instances(...)` in group_scored_batch (streaming.py:147-255): the fallback path reads it the same way
            max_instances = None
        info = h["info"]
        eff = info.eff_scale.detach().to("cpu", torch.float32).contiguous()
        edges = self.__dict__.get("_edges_i32")
        if edges is None:
            edges = self.__dict__["_edges_i32"] = np.ascontiguousarray(np.asarray(list(sc.edge_inds), dtype=np.int32).reshape(-1, 2))
        out_cap = max(1, max_instances) if max_instances is not None else max(self.__dict__.get("_inst_cap", 16), 1)
        status = np.zeros(4, dtype=np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data)
        while True:
            kp = np.empty((B, out_cap, n_nodes, 2), dtype=np.float32)
            vals = np.empty((B, out_cap, n_nodes), dtype=np.float32)
            scores = np.empty((B, out_cap), dtype=np.float32)
            n_inst = np.zeros((B,), dtype=np.int32)
            rc = L.lib().ph_group_packed(
                C.c_void_p(h["buf"].data_ptr()), B, n_nodes, h["peak_cap"], h["cand_cap"], p(edges), edges.shape[0], float(sc.min_line_scores), float(sc.min_instance_peaks),
                1 if isinstance(sc.min_instance_peaks, float) else 0, -1 if max_instances is None else int(max_instances),
                -1 if self.max_peaks_per_node is None else int(self.max_peaks_per_node), float(info.input_scale), C.c_void_p(eff.data_ptr()), out_cap, p(kp), p(vals), p(scores),
                p(n_inst), p(status),
            )
            if rc == L.PH_E_INFEASIBLE:
                raise ValueError("cost matrix is infeasible")  # what scipy raises inside the reference
            L.check(rc)
            if status[2] & 1:  # rare: the arena's capacities were too small -> grow and redo this batch synchronously
                self._peak_cap = max(self._peak_cap, int(status[0] * 1.25) + 16)
                self._cand_cap = max(self._cand_cap, int(status[1] * 1.25) + 16)
                return self._finish_packed(self._redo_eagerly(h))
            if status[2] & 2:  # more peaks in a frame than the output holds rows for
                out_cap = self.__dict__["_inst_cap"] = int(status[3])
                continue
            break
        self._pinned.setdefault(h["key"], []).append(h["buf"])  # the arena was consumed: the pinned buffer can be reused
        self._peak_cap, self._cand_cap = max(self._peak_cap, h["peak_cap"]), max(self._cand_cap, h["cand_cap"])
        mi = int(status[3])
        return Outputs(pred_keypoints=torch.from_numpy(kp[:, :mi]).contiguous(), pred_peak_values=torch.from_numpy(vals[:, :mi]).contiguous(),
                       instance_scores=torch.from_numpy(scores[:, :mi]).contiguous(), preprocess_info=info.cpu())

    def _finish_scoring(self, h: dict) -> ScoredBatch:
        pc0 = self.postprocess_config
        if h["raw"] is None and (pc0.return_confmaps or pc0.return_pafs or pc0.return_paf_graph):  # a graphed handle holds no head tensors to hand back
            h = self._redo_eagerly(h)
        h["event"].synchronize()
        B, n_nodes, pcap, ccap = h["B"], h["n_nodes"], h["peak_cap"], h["cand_cap"]
        arr = h["buf"].numpy()
        head = arr[: h["n_head"]].view(np.int32)
        n_peaks = int(head[0])
        n_cand = int(head[2 + 2 * B + B])
        if n_peaks > pcap or n_cand > ccap:  # rare: capacity too small -> grow and redo this batch synchronously
            self._peak_cap = max(self._peak_cap, int(n_peaks * 1.25) + 16)
            self._cand_cap = max(self._cand_cap, int(n_cand * 1.25) + 16)
            return self._finish_scoring(self._redo_eagerly(h))
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
        skip = False
        if self.max_peaks_per_node is not None:
            for b in range(B):
                c = peak_channel[peak_offsets[b] : peak_offsets[b + 1]]
                if c.size and int(np.bincount(c, minlength=n_nodes).max()) > self.max_peaks_per_node:
                    skip = True  # combinatorial-blow-up guard (bottomup.py:126-161)
                    break
        keep_cms = pc.return_confmaps or (pc.return_paf_graph and not skip)
        keep_pafs = pc.return_pafs or (pc.return_paf_graph and not skip)
        cms = h["raw"]["MultiInstanceConfmapsHead"] if keep_cms else None
        pafs = h["raw"]["PartAffinityFieldsHead"] if keep_pafs else None
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
