"""bench.py leg: the one workload the reference publishes numbers for (leg `published_workload`)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from benchlegs.common import *  # noqa: F401,F403  (constants + helpers; the names are listed in common.__all__)
from benchlegs.common import ROOT, _cfg5_traffic, _forward_profile, _matrix_rows, _pad16, _small_roofline, _time_calls  # noqa: F401


# docs/guides/inference-performance.md:40-48,70-77: the only numbers the reference publishes, NVIDIA A40 / CUDA 12.8 / torch 2.9.1
PUBLISHED_A40 = {"bottomup_forward_ms_per_batch4": {"eager_fp32": 3.59, "torch_compile": 2.94, "fp16_autocast": 2.32}, "bottomup_end_to_end_fps": 137.0,
                 "single_instance_forward_ms_per_batch4": {"eager_fp32": 1.20, "torch_compile": 0.93, "fp16_autocast": 0.84}, "single_instance_end_to_end_fps": 228.0,
                 "centroid_forward_ms_per_batch4": {"eager_fp32": 2.48, "torch_compile": 1.96, "fp16_autocast": 1.61}, "topdown_end_to_end_fps": 95.0}


PUBLISHED_LANES = 3  # copies of a small network Predictor keeps in flight on as many HIP streams (Predictor.from_model_paths(streams=...)' default)


def published_workload_leg(steps, dev):
    """The one workload the reference publishes numbers for (docs/guides/inference-performance.md:40-48,70-77, an NVIDIA A40): its fixture bottom-up run directory (tests/golden/ckpt_dirs:
    UNet f16 / rate 1.5 / max_stride 8, transposed-conv decoder, 2 nodes / 1 edge) at 320 x 560, batch 4 (predictor.py:884,930).  Backbone-level forward per batch (their table 1) in exact fp32
    and in the autocast-equivalent fp16 mode, and end-to-end frames/s of Predictor.predict over 100 frames (their table 2; theirs includes video decoding, ours starts from uint8 frames in host memory)."""
    from sleap_nn_amd.inference.backends import HipBackend
    from sleap_nn_amd.inference.loaders import load_model_assets
    from sleap_nn_amd.inference.predictor import Predictor

    root = os.path.join(ROOT, "tests", "golden", "ckpt_dirs", "minimal_instance_bottomup")
    model = load_model_assets(root).build_model().to(dev)
    g = torch.Generator().manual_seed(4321)
    frames = torch.randint(0, 256, (4, 1, 320, 560), dtype=torch.uint8, generator=g).to(dev)
    table, op_ms, codes, executed, direct, matrix_ms, kernels = _forward_profile(model, frames)
    fwd = {}
    for tag, kw in (("exact_fp32", {}), ("fp16_autocast_equivalent", {"use_fp16": True})):
        backend = HipBackend(model, str(dev), use_graph=True, **kw)
        x5 = backend.static_input(tuple(frames.shape)).copy_(frames)
        n = max(steps, 200)
        total, _ = _time_calls(lambda: backend(x5), n, 20, False)
        fwd[tag] = 1e3 * total / n
    model.set_precision("exact")
    fwd_s = fwd["exact_fp32"] * 1e-3
    # end to end: Predictor on 100 host frames (real texture: the fixture video's two golden frames tiled to 320 x 560), batch 4
    z = np.load(os.path.join(ROOT, "tests", "golden", "ckpt_bottomup.npz"), allow_pickle=False)
    two = torch.from_numpy(z["image"]).squeeze(1)
    vid = torch.cat([two, two.flip(-1)], 0)[:, :, 32:352, :].repeat(25, 1, 1, 2)[..., :560].contiguous()  # (100, 1, 320, 560) uint8, host
    pred = Predictor.from_model_paths([root], device=str(dev), batch_size=4, peak_threshold=0.2, streams=PUBLISHED_LANES)
    pred.predict(vid)  # (untimed: graph capture of the batch shape, pinned buffers, the host-stage worker)
    torch.cuda.synchronize()
    reps, t0 = 10, time.perf_counter()
    n_inst = 0
    for _ in range(reps):
        outs = pred.predict(vid)
        n_inst = sum(int((~torch.isnan(o.instance_scores)).sum()) for o in outs)
    e2e = reps * vid.shape[0] / (time.perf_counter() - t0)
    ref = PUBLISHED_A40
    # the same two measurements for the reference's single-instance fixture (its run directory carries input scale 0.5: the 320 x 560 frames reach the backbone as 160 x 280)
    si_root = os.path.join(ROOT, "tests", "golden", "ckpt_dirs", "minimal_instance_single_instance")
    si = {}
    try:
        si_model = load_model_assets(si_root).build_model().to(dev)
        si_frames = torch.randint(0, 256, (4, 1, 160, 280), dtype=torch.uint8, generator=g).to(dev)
        for tag, kw in (("exact_fp32", {}), ("fp16_autocast_equivalent", {"use_fp16": True})):
            be = HipBackend(si_model, str(dev), use_graph=True, **kw)
            xb = be.static_input(tuple(si_frames.shape)).copy_(si_frames)
            n = max(steps, 200)
            tot, _ = _time_calls(lambda: be(xb), n, 20, False)
            si.setdefault("forward_ms_per_batch", {})[tag] = 1e3 * tot / n
        si_model.set_precision("exact")
        sp = Predictor.from_model_paths([si_root], device=str(dev), batch_size=4, peak_threshold=0.2, streams=PUBLISHED_LANES)
        sp.predict(vid)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            souts = sp.predict(vid)
            souts[-1].pred_keypoints.cpu()
        si["end_to_end_fps"] = reps * vid.shape[0] / (time.perf_counter() - t1)
        si["vs_baseline"] = {"forward_eager_fp32": ref["single_instance_forward_ms_per_batch4"]["eager_fp32"] / si["forward_ms_per_batch"]["exact_fp32"],
                             "forward_fp16": ref["single_instance_forward_ms_per_batch4"]["fp16_autocast"] / si["forward_ms_per_batch"]["fp16_autocast_equivalent"],
                             "end_to_end_fps": si["end_to_end_fps"] / ref["single_instance_end_to_end_fps"]}
        si["what"] = "fixture single-instance run directory: forward on (4, 1, 160, 280) (input scale 0.5 of the 320 x 560 frames), Predictor.predict over the same 100 host frames (antialiased resize + forward + global peaks)"
    except Exception as e:  # the headline legs must not die on the extra fixture
        si = {"error": repr(e)}
    # two-stage top-down (centroid -> crops -> centered instance) on the reference's fixture models (tests/golden/topdown.npz holds their weights and configs)
    td = {}
    try:
        from sleap_nn_amd.architectures.model import Model
        from sleap_nn_amd.inference.layers import CenteredInstanceLayer, CentroidLayer, PostprocessConfig, TopDownLayer

        tz = np.load(os.path.join(ROOT, "tests", "golden", "topdown.npz"), allow_pickle=False)
        tcfg = json.loads(str(tz["config_json"]))
        cc, ci = tcfg["centroid"], tcfg["centered"]
        wsel = lambda pre: {k[len(pre):]: torch.from_numpy(tz[k]) for k in tz.files if k.startswith(pre)}
        mc = Model("unet", cc["backbone"], cc["heads"], "centroid")
        mc.load_state_dict(wsel("wc/"))
        mi = Model("unet", ci["backbone"], ci["heads"], "centered_instance")
        mi.load_state_dict(wsel("wi/"))
        def make_tdl(mc_, mi_):
            cbe_ = HipBackend(mc_, str(dev), use_graph=True)
            cl_ = CentroidLayer(cbe_, cc["heads"]["confmaps"]["output_stride"], max_instances=6, max_stride=cc["backbone"]["max_stride"], postprocess_config=PostprocessConfig(peak_threshold=0.03, max_instances=6))
            il_ = CenteredInstanceLayer(HipBackend(mi_, str(dev)), ci["heads"]["confmaps"]["output_stride"], max_stride=ci["backbone"]["max_stride"], postprocess_config=PostprocessConfig(peak_threshold=0.03))
            return cbe_, TopDownLayer(cl_, il_, (tcfg["crop_size"], tcfg["crop_size"]))

        cbe, tdl = make_tdl(mc, mi)
        td_replicas = []
        for _ in range(PUBLISHED_LANES - 1):  # further copies of the pair (own handles): Predictor's other lanes
            mc2 = Model("unet", cc["backbone"], cc["heads"], "centroid")
            mc2.load_state_dict(wsel("wc/"))
            mi2 = Model("unet", ci["backbone"], ci["heads"], "centered_instance")
            mi2.load_state_dict(wsel("wi/"))
            td_replicas.append(make_tdl(mc2, mi2)[1])
        tframes = torch.from_numpy(tz["image"]).to(dev)
        tframes = tframes.reshape(-1, *tframes.shape[-3:])
        tframes = tframes.repeat((4 + tframes.shape[0] - 1) // tframes.shape[0], 1, 1, 1)[:4].contiguous()
        cx = cbe.static_input(tuple(tframes.shape)).copy_(tframes)
        n = max(steps, 200)
        tot, _ = _time_calls(lambda: cbe(cx), n, 20, False)
        td["centroid_forward_ms_per_batch"] = 1e3 * tot / n
        tout = tdl.predict(tframes)
        tot, _ = _time_calls(lambda: tdl.predict(tframes), 100, 10, False)
        td["layer_predict_ms_per_batch"] = 1e3 * tot / 100
        td["layer_predict_fps"] = 4 * 100 / tot
        td["instances_per_batch"] = int(torch.isfinite(tout.pred_centroids[..., 0]).sum())
        # end to end as for the bottom-up model: Predictor.predict over 100 uint8 frames in host memory, batch 4 (stage 1 of batch i + 1 enqueued before the one host read of batch i)
        tvid = tframes.cpu().repeat(25, 1, 1, 1).contiguous()
        tp = Predictor(tdl, batch_size=4, replicas=td_replicas)
        tp.predict(tvid)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(reps):
            touts = tp.predict(tvid)
            touts[-1].pred_keypoints.cpu()
        tot = time.perf_counter() - t2
        td["end_to_end_ms_per_batch"] = 1e3 * tot / (reps * 25)
        td["end_to_end_fps"] = reps * tvid.shape[0] / tot
        td["vs_baseline"] = {"centroid_forward_eager_fp32": ref["centroid_forward_ms_per_batch4"]["eager_fp32"] / td["centroid_forward_ms_per_batch"], "end_to_end_fps": td["end_to_end_fps"] / ref["topdown_end_to_end_fps"]}
        td["what"] = f"fixture top-down models (tests/golden/topdown.npz): {tuple(tframes.shape)} uint8 frames resident in HBM, centroid forward (hipGraph) and TopDownLayer.predict (centroid -> NMS peaks -> device-side selection -> {tcfg['crop_size']} x {tcfg['crop_size']} crops -> centered-instance forward -> global peaks -> scatter; ONE host read per batch: the per-frame centroid counts); end_to_end = Predictor.predict over 100 host frames, pipelined over that read, two copies of the layer pair on two HIP streams"
    except Exception as e:
        td = {"error": repr(e)}
    return {"metric": "ms per batch of 4, bottom-up backbone forward (the reference's published table)", "value": fwd["exact_fp32"], "unit": "ms/batch", "higher_is_better": False, "steps": max(steps, 200),
            "dtype": "f32", "data": "reference fixture checkpoint (tests/golden/ckpt_dirs/minimal_instance_bottomup), synthetic uint8 frames",
            "config": {"workload": "published: fixture bottom-up UNet (f16, rate 1.5, max_stride 8, transposed-conv decoder, 2 nodes / 1 edge), 320x560x1 uint8, batch 4", "frames_per_step": 4,
                       "params": model.num_parameters(), "forward_launch": "hipGraph replay, back to back"},
            "forward_ms_per_batch": fwd, "frames_per_s_forward": 4.0 / fwd_s,
            "end_to_end": {"value": e2e, "unit": "frames/s", "frames": int(vid.shape[0]), "repeats": reps, "instances_found_per_pass": n_inst,
                           "what": "Predictor.predict (pipelined: pinned staging + H2D, resize / pad + forward + peaks + PAF scoring as one hipGraph, D2H, one-call C++ grouping in a worker; consecutive batches alternate between three copies of the layer on three HIP streams, Predictor.from_model_paths(streams=3)) over 100 uint8 frames in host memory, batch 4, exact fp32"},
            "vs_baseline": {"forward_eager_fp32": ref["bottomup_forward_ms_per_batch4"]["eager_fp32"] / fwd["exact_fp32"], "forward_fp16": ref["bottomup_forward_ms_per_batch4"]["fp16_autocast"] / fwd["fp16_autocast_equivalent"],
                            "end_to_end_fps": e2e / ref["bottomup_end_to_end_fps"], "reference": ref, "reference_hardware": "NVIDIA A40, CUDA 12.8, torch 2.9.1 (docs/guides/inference-performance.md:3-7,40-48,70-77)",
                            "note": "ratios > 1 = this build faster; different hardware and (end to end) no video decoding here: a like-for-like of the workload, not of the machine"},
            "single_instance": si, "topdown": td,
            "roofline": _small_roofline(executed, direct, matrix_ms, fwd_s, kernels, len(table))}
