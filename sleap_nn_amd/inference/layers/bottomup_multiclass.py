"""``BottomUpMultiClassLayer`` (sleap_nn/inference/layers/bottomup_multiclass.py:26-190)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from sleap_nn_amd.inference.backends import ModelBackend
from sleap_nn_amd.inference.layers.base import InferenceLayer
from sleap_nn_amd.inference.layers.configs import PostprocessConfig, PreprocessConfig
from sleap_nn_amd.inference.outputs import Outputs
from sleap_nn_amd.inference.preprocess_info import PreprocInfo


class BottomUpMultiClassLayer(InferenceLayer):
    def __init__(self, backend: ModelBackend, cms_output_stride: int, class_maps_output_stride: int, max_instances: Optional[int] = None,
                 max_stride: int = 1, preprocess_config: Optional[PreprocessConfig] = None, postprocess_config: Optional[PostprocessConfig] = None) -> None:
        super().__init__(backend, preprocess_config or PreprocessConfig(), postprocess_config or PostprocessConfig(), cms_output_stride, max_stride)
        self.cms_output_stride = cms_output_stride
        self.class_maps_output_stride = class_maps_output_stride
        self.max_instances = max_instances

    def postprocess(self, raw_out: dict, info: PreprocInfo) -> Outputs:
        return self._finish_postprocess(self._enqueue_postprocess(raw_out, info))

    def _enqueue_postprocess(self, raw_out: dict, info: PreprocInfo) -> dict:
        """The GPU stage -- peak finding, class-map sampling at every peak -- enqueued without a host read, its results copied asynchronously into ONE pinned arena
        ``[counts 2 + 2B | xy 2P | vals P | class probabilities P K | sample P | channel P]``; ``_finish_postprocess`` (any thread) waits for the event and does the host stage
        (Hungarian matching per (sample, node), scatter by class: bottomup_multiclass.py:76-150).  A caller that enqueues the next batch before finishing this one keeps the GPU busy."""
        import ctypes as C

        from sleap_nn_amd import _lib as L
        from sleap_nn_amd.inference.ops.peaks import find_local_peaks_device

        cms = raw_out["MultiInstanceConfmapsHead"]
        class_maps = raw_out["ClassMapsHead"].detach().to(torch.float32).contiguous()
        pc = self.postprocess_config
        B, n_nodes = int(cms.shape[0]), int(cms.shape[1])
        K = int(class_maps.shape[1])
        dev = cms.device
        cap = max(self.__dict__.get("_peak_cap", 0), B * n_nodes * 16, 1024)
        n_head = 2 + 2 * B
        packed = torch.empty(n_head + (5 + K) * cap, dtype=torch.float32, device=dev)
        ints = packed.view(torch.int32)
        o = n_head
        xy = packed[o : o + 2 * cap].view(cap, 2)
        o += 2 * cap
        vals = packed[o : o + cap]
        o += cap
        probs = packed[o : o + cap * K].view(cap, K)
        o += cap * K
        sb, sc = ints[o : o + cap], ints[o + cap : o + 2 * cap]
        counts = ints[:n_head]
        # peaks * cms_output_stride (bottomup_multiclass.py) is the kernel's last multiply; the class maps are sampled at peaks / class_maps_output_stride
        _xy, _vals, sb_full, _sc, _counts, _ = find_local_peaks_device(cms, pc.peak_threshold, pc.effective_refinement, pc.integral_patch_size, cap,
                                                                       xy_scale=float(self.cms_output_stride), out=(xy, vals, sc, counts))
        sb.copy_(sb_full.clamp_(0, B - 1))  # (rows beyond the count are uninitialised: keep their sample index inside the tensor for the sampling kernel)
        q = xy / self.class_maps_output_stride
        with torch.cuda.device(dev):  # (rows beyond the count hold whatever the arena held: their samples are clamped reads that nobody looks at)
            L.check(L.lib().ph_sample_class_maps(C.c_void_p(class_maps.data_ptr()), B, K, int(class_maps.shape[2]), int(class_maps.shape[3]), C.c_void_p(q.data_ptr()),
                                                 C.c_void_p(sb.data_ptr()), cap, C.c_void_p(probs.data_ptr()), L.current_stream_ptr()))
        xy.copy_(q)  # (the arena carries the class-map coordinates: classify_peaks_from_maps returns the points it was given)
        pool = self.__dict__.setdefault("_pinned", {})
        free = pool.setdefault(int(packed.numel()), [])
        buf = free.pop() if free else torch.empty(packed.numel(), dtype=torch.float32, pin_memory=True)
        buf.copy_(packed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        return {"buf": buf, "event": ev, "B": B, "n_nodes": n_nodes, "K": K, "cap": cap, "raw": raw_out, "info": info, "keep": packed}

    def _finish_postprocess(self, h: dict) -> Outputs:
        from sleap_nn_amd.inference.ops.identity import group_class_peaks

        h["event"].synchronize()
        B, n_nodes, K, cap = h["B"], h["n_nodes"], h["K"], h["cap"]
        arr = h["buf"]
        n = int(arr.view(torch.int32)[0])
        if n > cap:  # rare: more peaks than rows -> once more with room for all of them
            self.__dict__["_peak_cap"] = int(n * 1.25) + 16
            self._pinned[int(arr.numel())].append(arr)
            return self._finish_postprocess(self._enqueue_postprocess(h["raw"], h["info"]))
        n_head = 2 + 2 * B
        o = n_head
        pts = arr[o : o + 2 * cap].view(cap, 2)[:n].clone()
        o += 2 * cap
        vals = arr[o : o + cap][:n].clone()
        o += cap
        probs = arr[o : o + cap * K].view(cap, K)[:n].clone()
        o += cap * K
        ints = arr.view(torch.int32)
        sb = ints[o : o + cap][:n].long()
        sc = ints[o + cap : o + 2 * cap][:n].long()
        self._pinned[int(arr.numel())].append(arr)
        info, pc = h["info"], self.postprocess_config
        cms, class_maps = h["raw"]["MultiInstanceConfmapsHead"], h["raw"]["ClassMapsHead"]
        # classify_peaks_from_maps on the host copies (identity.py:79-146)
        pi, ci = group_class_peaks(probs, sb, sc, B, n_nodes)
        inst = torch.full((B, K, n_nodes, 2), float("nan"))
        pvals = torch.full((B, K, n_nodes), float("nan"))
        cprobs = torch.full((B, K, n_nodes), float("nan"))
        inst[sb[pi], ci, sc[pi]] = pts[pi]
        pvals[sb[pi], ci, sc[pi]] = vals[pi]
        cprobs[sb[pi], ci, sc[pi]] = probs[pi, ci]
        inst = inst * self.class_maps_output_stride
        if info.input_scale != 1.0:
            inst = inst / info.input_scale
        eff = info.eff_scale.cpu()
        if not torch.all(eff == 1.0):
            inst = inst / eff.view(-1, 1, 1, 1)
        iscores = torch.nanmean(pvals, dim=-1)
        tscores = torch.nanmean(cprobs, dim=-1)
        max_instances = getattr(pc, "max_instances", None) or self.max_instances
        if max_instances is not None:
            for b in range(inst.shape[0]):  # keep the top-N classes by score, in place (bottomup_multiclass.py:152-190)
                s = iscores[b]
                if int((~torch.isnan(s)).sum()) <= max_instances:
                    continue
                order = np.argsort(s.numpy())[::-1]
                drop = torch.ones(s.shape[0], dtype=torch.bool)
                drop[torch.as_tensor(order[:max_instances].copy(), dtype=torch.long)] = False
                inst[b, drop], pvals[b, drop], iscores[b, drop], tscores[b, drop] = float("nan"), float("nan"), float("nan"), float("nan")
        out = Outputs(pred_keypoints=inst, pred_peak_values=pvals, instance_scores=iscores, preprocess_info=info)
        out.instance_tracking_scores = tscores
        if pc.return_confmaps:
            out.pred_confmaps = cms.detach()
        if pc.return_class_maps:
            out.pred_class_maps = class_maps.detach()
        return out
