"""``TopDownLayer`` (sleap_nn/inference/layers/topdown.py:36-466): centroids -> crops of the
full-resolution frame (GPU gather) -> centered-instance peaks -> image coordinates."""
from __future__ import annotations

from typing import Tuple

import torch

from sleap_nn_amd.inference.layers.centered_instance import CenteredInstanceLayer
from sleap_nn_amd.inference.layers.centroid import CentroidLayer
from sleap_nn_amd.inference.ops.coord import add_crop_offset
from sleap_nn_amd.inference.ops.crops import crop_bboxes, make_centered_bboxes
from sleap_nn_amd.inference.outputs import Outputs


class TopDownLayer:
    def __init__(self, centroid_layer: CentroidLayer, centered_instance_layer: CenteredInstanceLayer, crop_size: Tuple[int, int],
                 centroid_nms: bool = False, centroid_nms_threshold: float = 0.5, return_crops: bool = False) -> None:
        self.centroid_nms = bool(centroid_nms)
        self.centroid_nms_threshold = float(centroid_nms_threshold)
        self.centroid_layer = centroid_layer
        self.centered_instance_layer = centered_instance_layer
        self.crop_size = crop_size
        self.return_crops = return_crops

    def predict(self, image) -> Outputs:
        return self._finish(self._enqueue_stage1(image))

    def _enqueue_stage1(self, image) -> dict:
        """Stage 1 of a batch -- preprocessing, centroid forward, peak finding -- enqueued without a host read, plus the asynchronous D2H of the per-frame peak counts.
        ``_finish`` reads those counts (the ONE host sync of a batch) and runs the rest; a caller that enqueues the next batch's stage 1 before finishing this one
        (``Predictor.predict``) keeps the GPU busy across that sync."""
        cl = self.centroid_layer
        x = cl._to_4d_tensor(image).to(torch.device(cl.backend.device), non_blocking=True)
        if self.centroid_nms:
            return {"x": x, "slow": True}
        xp, info = cl.preprocess(x)
        raw = cl.backend(xp)
        return {"x": x, "slow": False, "raw": raw, "info": info, "sel": cl._select_enqueue(raw, info, frames=xp)}

    def _sized_frames(self, x: torch.Tensor, info) -> torch.Tensor:
        """The frames stage 2 is cut from (layers/topdown.py:127-150, ``_sizematch_like_centroid_layer``): the RAW frames through the centroid layer's sizematcher
        -- no channel coercion, no input scale, no pad to stride.  The centered-instance model was trained on crops of sized frames."""
        cfg = self.centroid_layer.preprocess_config
        if cfg.max_height is None and cfg.max_width is None:
            return x
        if (cfg.max_height is None or cfg.max_height == x.shape[-2]) and (cfg.max_width is None or cfg.max_width == x.shape[-1]):
            return x
        from sleap_nn_amd.data.resizing import apply_sizematcher

        sized, _e = apply_sizematcher(x, cfg.max_height, cfg.max_width)
        return sized

    def _finish(self, h: dict) -> Outputs:
        import ctypes as C

        from sleap_nn_amd import _lib as L

        if h["slow"]:
            return self._predict_with_host_nms(h["x"])
        cl, il = self.centroid_layer, self.centered_instance_layer
        x = h["x"]
        dev = x.device
        ch, cw = self.crop_size
        sel = cl._select_finish(h["sel"], h["info"], (ch, cw))
        centroids, cvals, I, n_valid = sel["centroids"], sel["vals"], sel["I"], sel["n_valid"]
        B = int(centroids.shape[0])
        if n_valid == 0:
            # (layers/topdown.py:213-230: all-NaN keypoints of the right shape, on the centroid model's device, like every other batch of the video)
            n_nodes = self._infer_n_nodes()
            out = Outputs(pred_keypoints=torch.full((B, I, n_nodes, 2), float("nan"), device=dev), pred_peak_values=torch.full((B, I, n_nodes), float("nan"), device=dev),
                          pred_centroids=centroids, pred_centroid_values=cvals, instance_scores=cvals, preprocess_info=h["info"])
            out.pred_crop_keypoints = torch.full((B, I, n_nodes, 2), float("nan"), device=dev)
            out.instance_bboxes = torch.full((B, I, 4, 2), float("nan"), device=dev)
            return out
        x = self._sized_frames(x, h["info"])
        if x.dtype == torch.uint8:
            code = 0
        elif x.dtype == torch.float32:
            code = 1
        else:
            raise TypeError(f"crop_bboxes supports uint8 and float32 images, got {x.dtype}")
        x = x.contiguous()
        _b, Cc, H, W = x.shape
        crops = torch.empty((n_valid, Cc, ch, cw), dtype=x.dtype, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().ph_crop_bboxes(C.c_void_p(x.data_ptr()), code, B, Cc, H, W, C.c_void_p(sel["list_tl"].data_ptr()), C.c_void_p(sel["list_sample"].data_ptr()), n_valid,
                                           ch, cw, C.c_void_p(crops.data_ptr()), L.current_stream_ptr()))
        s2 = il.predict(crops)
        k3 = s2.pred_keypoints.squeeze(1).contiguous()
        v3 = s2.pred_peak_values.squeeze(1).contiguous()
        n_nodes = int(k3.shape[-2])
        full_k = torch.empty((B, I, n_nodes, 2), dtype=torch.float32, device=dev)
        full_c = torch.empty((B, I, n_nodes, 2), dtype=torch.float32, device=dev)
        full_v = torch.empty((B, I, n_nodes), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().ph_topdown_scatter(C.c_void_p(k3.data_ptr()), C.c_void_p(v3.data_ptr()), C.c_void_p(sel["list_tl"].data_ptr()), C.c_void_p(sel["pos_of_slot"].data_ptr()),
                                               B * I, n_nodes, C.c_void_p(sel["eff"].data_ptr()), I, C.c_void_p(full_k.data_ptr()), C.c_void_p(full_c.data_ptr()),
                                               C.c_void_p(full_v.data_ptr()), L.current_stream_ptr()))
        out = Outputs(pred_keypoints=full_k, pred_crop_keypoints=full_c, pred_peak_values=full_v, pred_centroids=centroids, pred_centroid_values=cvals,
                      instance_scores=cvals, preprocess_info=h["info"])
        out.instance_bboxes = sel["bboxes"]
        if s2.pred_class_probs is not None or self.return_crops:
            slots = sel["list_slot"][:n_valid].long()
            idx = torch.stack([slots // I, slots % I], dim=1)
            self._attach_identity_and_crops(out, s2, idx, crops, B, I, n_nodes)
        return out

    def _infer_n_nodes(self) -> int:
        """Node count of the centered-instance model (its confidence-map head's channels), for the all-NaN outputs of a batch without centroids."""
        model = getattr(self.centered_instance_layer.backend, "model", None)
        for head in getattr(model, "heads", None) or []:
            if type(head).__name__ == "CenteredInstanceConfmapsHead":
                return int(head.channels)
        return 1

    def _predict_with_host_nms(self, x: torch.Tensor) -> Outputs:
        """The path with centroid NMS (layers/topdown.py:395-438: host logic over the centroids of a frame): centroids -> host -> mask -> crops, as the reference walks it."""
        cout = self.centroid_layer.predict(x)
        centroids, cvals = cout.pred_centroids, cout.pred_centroid_values
        B, I, _ = centroids.shape
        dev = centroids.device
        valid = ~torch.isnan(centroids).any(dim=-1)
        valid = valid & self._centroid_nms_mask(centroids, cvals, valid)
        idx = valid.nonzero(as_tuple=False)
        n_valid = int(idx.shape[0])
        ch, cw = self.crop_size
        info = cout.preprocess_info
        if n_valid == 0:
            n_nodes = self._infer_n_nodes()
            out = Outputs(pred_keypoints=torch.full((B, I, n_nodes, 2), float("nan"), device=dev), pred_peak_values=torch.full((B, I, n_nodes), float("nan"), device=dev),
                          pred_centroids=centroids, pred_centroid_values=cvals, instance_scores=cvals, preprocess_info=info)
            out.pred_crop_keypoints = torch.full((B, I, n_nodes, 2), float("nan"), device=dev)
            out.instance_bboxes = torch.full((B, I, 4, 2), float("nan"), device=dev)
            return out
        eff = info.eff_scale.to(dev, torch.float32)
        per_crop_eff = eff[idx[:, 0]].view(-1, 1, 1)
        vc = centroids[idx[:, 0], idx[:, 1]] * per_crop_eff.view(-1, 1)  # sized space (layers/topdown.py:147)
        bboxes = make_centered_bboxes(vc, ch, cw)
        crops = crop_bboxes(self._sized_frames(x, info), bboxes, idx[:, 0])
        s2 = self.centered_instance_layer.predict(crops)
        k3 = s2.pred_keypoints.squeeze(1)
        kimg = add_crop_offset(k3, bboxes[:, 0, :]) / per_crop_eff
        bboxes = bboxes / per_crop_eff
        n_nodes = kimg.shape[-2]
        full_k = torch.full((B, I, n_nodes, 2), float("nan"), device=dev)
        full_c = torch.full((B, I, n_nodes, 2), float("nan"), device=dev)
        full_v = torch.full((B, I, n_nodes), float("nan"), device=dev)
        full_b = torch.full((B, I, 4, 2), float("nan"), device=dev)
        full_k[idx[:, 0], idx[:, 1]] = kimg
        full_c[idx[:, 0], idx[:, 1]] = k3
        full_v[idx[:, 0], idx[:, 1]] = s2.pred_peak_values.squeeze(1)
        full_b[idx[:, 0], idx[:, 1]] = bboxes
        out = Outputs(pred_keypoints=full_k, pred_crop_keypoints=full_c, pred_peak_values=full_v, pred_centroids=centroids, pred_centroid_values=cvals,
                      instance_scores=cvals, preprocess_info=cout.preprocess_info)
        out.instance_bboxes = full_b
        self._attach_identity_and_crops(out, s2, idx, crops, B, I, n_nodes)
        return out

    def _attach_identity_and_crops(self, out: Outputs, s2: Outputs, idx: torch.Tensor, crops: torch.Tensor, B: int, I: int, n_nodes: int) -> None:
        dev = idx.device
        ch, cw = self.crop_size
        if s2.pred_class_probs is not None:
            # multi-class identity (layers/topdown.py:333-390): classify the crops of each frame on their own
            from sleap_nn_amd.inference.ops.identity import get_class_inds_from_vectors

            vecs = s2.pred_class_probs.squeeze(1)
            full_ci = torch.full((B, I, n_nodes), -1, dtype=torch.int64, device=dev)
            full_ts = torch.full((B, I), float("nan"), device=dev)
            want = getattr(getattr(self.centered_instance_layer, "postprocess_config", None), "return_class_vectors", False)
            full_cv = torch.full((B, I, vecs.shape[-1]), float("nan"), device=dev) if want else None
            for b in torch.unique(idx[:, 0]).tolist():
                rows = (idx[:, 0] == b).nonzero(as_tuple=False).flatten()
                ci, cp = get_class_inds_from_vectors(vecs[rows])
                slots = idx[rows, 1]
                full_ci[b, slots] = ci.to(dev).view(-1, 1).expand(-1, n_nodes)
                full_ts[b, slots] = cp.to(dev)
                if full_cv is not None:
                    full_cv[b, slots] = vecs[rows]
            out.pred_class_inds, out.instance_tracking_scores, out.pred_class_vectors = full_ci, full_ts, full_cv
        if self.return_crops:
            fc = torch.zeros((B, I, crops.shape[1], ch, cw), dtype=crops.dtype, device=dev)
            fc[idx[:, 0], idx[:, 1]] = crops
            out.crops = fc

    def _centroid_nms_mask(self, centroids: torch.Tensor, centroid_vals: torch.Tensor, valid_mask: torch.Tensor) -> torch.Tensor:
        """Greedy NMS on the IoU of the crop boxes centred on each centroid (layers/topdown.py:395-438): per
        frame, in order of decreasing confidence, a centroid is dropped when its box overlaps an already
        kept one by more than ``centroid_nms_threshold``.  Host logic in fp32 (a handful of boxes per frame)."""
        import numpy as np

        c = centroids.detach().cpu().numpy().astype(np.float32)
        v = centroid_vals.detach().cpu().numpy().astype(np.float32)
        vm = valid_mask.detach().cpu().numpy()
        keep = np.ones_like(vm)
        h, w = np.float32(self.crop_size[0]), np.float32(self.crop_size[1])
        hh, hw = np.float32(self.crop_size[0] / 2.0), np.float32(self.crop_size[1] / 2.0)

        def iou(a, b):
            ih = max(np.float32(min(a[1] + hh, b[1] + hh) - max(a[1] - hh, b[1] - hh)), np.float32(0))
            iw = max(np.float32(min(a[0] + hw, b[0] + hw) - max(a[0] - hw, b[0] - hw)), np.float32(0))
            inter = np.float32(ih * iw)
            return inter / (np.float32(2.0) * (h * w) - inter)

        for b in range(c.shape[0]):
            vb = np.nonzero(vm[b])[0]
            if len(vb) <= 1:
                continue
            order = torch.from_numpy(v[b, vb]).argsort(descending=True).numpy()  # torch's ordering for ties, like the reference
            kept = []
            for j in order:
                cj = c[b, vb[j]]
                if any(iou(cj, k) > np.float32(self.centroid_nms_threshold) for k in kept):
                    keep[b, vb[j]] = False
                    continue
                kept.append(cj)
        return torch.from_numpy(keep).to(valid_mask.device)

    __call__ = predict
