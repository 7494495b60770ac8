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
        cout = self.centroid_layer.predict(image)
        centroids, cvals = cout.pred_centroids, cout.pred_centroid_values
        if centroids is None:
            return Outputs()
        B, I, _ = centroids.shape
        dev = centroids.device
        valid = ~torch.isnan(centroids).any(dim=-1)
        if self.centroid_nms:
            valid = valid & self._centroid_nms_mask(centroids, cvals, valid)
        idx = valid.nonzero(as_tuple=False)
        x = self.centroid_layer._to_4d_tensor(image).to(dev)
        n_valid = int(idx.shape[0])
        ch, cw = self.crop_size
        if n_valid == 0:
            n_nodes = 1
            return Outputs(pred_keypoints=torch.full((B, I, n_nodes, 2), float("nan")), pred_peak_values=torch.full((B, I, n_nodes), float("nan")),
                           pred_centroids=centroids.cpu(), pred_centroid_values=cvals.cpu(), instance_scores=cvals.cpu())
        vc = centroids[idx[:, 0], idx[:, 1]]
        bboxes = make_centered_bboxes(vc, ch, cw)
        crops = crop_bboxes(x, bboxes, idx[:, 0])
        s2 = self.centered_instance_layer.predict(crops)
        k3 = s2.pred_keypoints.squeeze(1)
        kimg = add_crop_offset(k3, bboxes[:, 0, :])
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
        return out

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
