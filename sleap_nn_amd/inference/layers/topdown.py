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
        if centroid_nms:
            raise NotImplementedError("centroid_nms is not part of the MI355X hot path yet")
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

    __call__ = predict
