"""``CentroidLayer`` (sleap_nn/inference/layers/centroid.py:44-271): centroid confmap -> local
peaks -> per-frame top-k, NaN-padded ``(B, I, 2)``."""
from __future__ import annotations

from typing import Optional

import torch

from sleap_nn_amd.inference.backends import ModelBackend
from sleap_nn_amd.inference.layers.base import InferenceLayer
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig
from sleap_nn_amd.inference.ops.peaks import find_local_peaks_device
from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo


class CentroidLayer(InferenceLayer):
    _HEAD_OUTPUT_KEY = "CentroidConfmapsHead"

    def __init__(self, backend: ModelBackend, output_stride: int, max_instances: Optional[int] = None, max_stride: int = 1,
                 anchor_ind: Optional[int] = None, use_gt_centroids: bool = False, preprocess_config: Optional[PreprocessConfig] = None,
                 postprocess_config: Optional[PostprocessConfig] = None) -> None:
        super().__init__(backend, preprocess_config or PreprocessConfig(), postprocess_config or PostprocessConfig(max_instances=max_instances), output_stride, max_stride)
        if use_gt_centroids:
            raise NotImplementedError("use_gt_centroids (LabelsReader path) is outside the MI355X hot path")
        self.max_instances = max_instances
        self.anchor_ind = anchor_ind
        self.use_gt_centroids = False

    def _select_enqueue(self, raw_out: dict, info: PreprocInfo, cap: int = 0, frames=None) -> dict:
        """Peak finding on the device (``ph_local_peaks``, coordinates x output stride) + the asynchronous D2H of its counts into pinned memory; no host sync."""
        cms = self._extract_confmaps(raw_out)
        pc = self.postprocess_config
        B = int(cms.shape[0])
        cap = max(1024, B * 256, cap)
        xy, vals, _sb, _sc, counts, cms_c = find_local_peaks_device(cms, pc.peak_threshold, pc.effective_refinement, pc.integral_patch_size, cap, xy_scale=float(info.output_stride))
        pool = self.__dict__.setdefault("_pinned_counts", {})
        free = pool.setdefault(B, [])
        host = free.pop() if free else torch.empty(2 + 2 * B, dtype=torch.int32, pin_memory=True)
        host.copy_(counts, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(cms.device))
        # `frames`: the preprocessed frames of this batch.  A graphed backend hands out its STATIC outputs, which the next batch's forward overwrites: the (rare) redo with more
        # peak rows in _select_finish then re-runs the forward on the frames instead of re-reading `raw_out` (BottomUpLayer._redo_eagerly does the same)
        return {"xy": xy, "vals": vals, "counts": counts, "counts_host": host, "event": ev, "B": B, "cap": cap, "raw": raw_out, "cms": cms_c, "frames": frames}

    def _select_finish(self, h: dict, info: PreprocInfo, crop_size=None, need_counts: bool = True) -> dict:
        """Per-frame selection on the device (``ph_centroid_select``).  Returns the padded centroids / values (device) and -- with ``crop_size`` -- the boxes and the
        stage-2 lists of ``TopDownLayer``.  ONE host read: the per-frame peak counts (``max_instances = None`` sizes the output by them, ``TopDownLayer`` the crop batch);
        with ``max_instances`` set and no lists wanted there is none (``need_counts=False``)."""
        import ctypes as C

        from sleap_nn_amd import _lib as L

        pc = self.postprocess_config
        B = h["B"]
        max_instances = getattr(pc, "max_instances", None) or self.max_instances
        lists = crop_size is not None
        per_frame = None
        if max_instances is None or lists or need_counts:
            h["event"].synchronize()  # the one sync of the stage
            counts_h = h["counts_host"]
            if int(counts_h[0]) > h["cap"]:  # rare: more peaks than rows -> once more with room for all of them
                raw = h["raw"]
                if h.get("frames") is not None and getattr(self.backend, "use_graph", False):
                    raw = self.backend(h["frames"])  # (the static outputs may already hold the next batch)
                h2 = self._select_enqueue(raw, info, cap=int(counts_h[0]), frames=h.get("frames"))
                return self._select_finish(h2, info, crop_size, need_counts)
            per_frame = counts_h[1 : 1 + B].clone()
            self._pinned_counts[B].append(counts_h)
        if max_instances is None:
            max_instances = int(per_frame.max()) if B > 0 else 0
        I = max(1, int(max_instances))
        xy, vals, counts = h["xy"], h["vals"], h["counts"]
        dev = xy.device
        cp = torch.empty((B, I, 2), dtype=torch.float32, device=dev)
        cv = torch.empty((B, I), dtype=torch.float32, device=dev)
        if bool((info.eff_scale == 1.0).all()):  # (the usual case: a cached device tensor, no H2D of B floats per batch)
            ones = self.__dict__.setdefault("_ones_dev", {})
            eff = ones.get((B, dev))
            if eff is None:
                eff = ones[(B, dev)] = torch.ones(B, dtype=torch.float32, device=dev)
        else:
            eff = info.eff_scale.to(dev, torch.float32).contiguous()
        res = {"centroids": cp, "vals": cv, "I": I, "eff": eff}
        if lists:
            res["bboxes"] = torch.empty((B, I, 4, 2), dtype=torch.float32, device=dev)
            res["list_sample"] = torch.empty((B * I,), dtype=torch.int32, device=dev)
            res["list_tl"] = torch.empty((B * I, 2), dtype=torch.float32, device=dev)
            res["list_slot"] = torch.empty((B * I,), dtype=torch.int32, device=dev)
            res["pos_of_slot"] = torch.empty((B * I,), dtype=torch.int32, device=dev)
            res["n_valid"] = int(torch.clamp(per_frame, max=I).sum())
        ptr = lambda k: C.c_void_p(res[k].data_ptr()) if k in res else None
        ch, cw = crop_size if lists else (0, 0)
        with torch.cuda.device(dev):
            L.check(L.lib().ph_centroid_select(
                C.c_void_p(xy.data_ptr()), C.c_void_p(vals.data_ptr()), C.c_void_p(counts.data_ptr()), B, I, int(xy.shape[0]), float(info.input_scale),
                C.c_void_p(eff.data_ptr()), float(ch), float(cw), C.c_void_p(cp.data_ptr()), C.c_void_p(cv.data_ptr()), ptr("bboxes"), ptr("list_sample"), ptr("list_tl"),
                ptr("list_slot"), ptr("pos_of_slot"), None, L.current_stream_ptr()))
        return res

    def postprocess(self, raw_out: dict, info: PreprocInfo) -> Outputs:
        sel = self._select_finish(self._select_enqueue(raw_out, info), info)
        out = Outputs(pred_centroids=sel["centroids"], pred_centroid_values=sel["vals"], preprocess_info=info)
        if self.postprocess_config.return_confmaps:
            out.pred_confmaps = self._extract_confmaps(raw_out).detach()
        return out
