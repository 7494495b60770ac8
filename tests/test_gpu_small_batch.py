"""GPU parity tests of the small-batch regime: BASELINE cfg1 (B = 1, 256 x 256), the reference's published workload (its fixture bottom-up run
directory at 320 x 560, batch 4: docs/guides/inference-performance.md:40-48) and the split-K form of the F(2x2,3x3) kernel that serves layers
with fewer (pixel tile, N tile) work units than CUs.  Same bars as tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O
from sleap_nn_amd import _lib as L
from tests import _golden as G

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
CMS_ATOL = 1e-4
HEAD_RTOL = 1e-5

SI_BB = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 16, "stem_stride": None, "middle_block": True, "up_interpolate": True,
         "stacks": 1, "convs_per_block": 2, "output_stride": 2}


def _close(got, ref, key=None, rtol=HEAD_RTOL):
    err = (got.cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= CMS_ATOL * max(1.0, scale) and err <= rtol * scale, (key, err, scale)


def test_cfg1_single_instance_256_batch1_network_and_global_peaks():
    """BASELINE cfg1 exactly: single-instance UNet f16/r2/max_stride 16/output_stride 2, 256 x 256 gray, 5 keypoints, ONE frame -- the batch at which
    every persistent kernel has fewer tiles than CUs and the K-heavy layers split K (default options: asserted from the kernel record).
    Network vs the oracle (1e-4 and 1e-5 of the head's scale), global peaks of the network's own maps vs the oracle's on the same maps (bit-exact
    values, 1e-4 px), a second identical launch bit-identical, and the same frame inside a batch of 3 within the relative bar."""
    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.inference.backends import HipBackend
    from sleap_nn_amd.inference.layers import SingleInstanceLayer
    from sleap_nn_amd.inference.preprocess_info import PreprocInfo

    heads = {"confmaps": {"part_names": [f"k{i}" for i in range(5)], "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0}}
    sd = O.init_state(SI_BB, heads, "single_instance", seed=11, head_scale=1.0)
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (3, 1, 256, 256), dtype=torch.uint8, generator=g)
    ref = O.model_forward(sd, SI_BB, heads, "single_instance", img[:1])["SingleInstanceConfmapsHead"]
    m = Model("unet", SI_BB, heads, "single_instance")
    m.load_state_dict(sd)
    for use_graph in (False, True):
        layer = SingleInstanceLayer(HipBackend(m, DEV, use_graph=use_graph), 2, max_stride=16)
        raw = layer.backend(img[:1])["SingleInstanceConfmapsHead"]
        assert tuple(raw.shape) == (1, 5, 128, 128)
        _close(raw, ref, "cfg1")
        first = raw.cpu().clone()
        assert torch.equal(layer.backend(img[:1])["SingleInstanceConfmapsHead"].cpu(), first)  # run-to-run bitwise (fixed-order second stage)
    # zero-copy input: the graph's own input buffer, filled by the caller, goes in without a staging copy -- and a refill is seen by the next replay
    buf = layer.backend.static_input((1, 1, 256, 256))
    buf.copy_(img[:1])
    assert torch.equal(layer.backend(buf)["SingleInstanceConfmapsHead"].cpu(), first)
    buf.copy_(img[1:2])
    other = layer.backend(buf)["SingleInstanceConfmapsHead"].cpu()
    assert not torch.equal(other, first) and torch.equal(other, layer.backend(img[1:2])["SingleInstanceConfmapsHead"].cpu())
    # the whole step as one graph (forward + global peaks + refinement + coordinate ladder): same bits as predict, and it follows a refill of its input buffer
    from sleap_nn_amd.inference.layers import PostprocessConfig

    layer.postprocess_config = PostprocessConfig(peak_threshold=-1.0)  # (random-weight maps sit below the default threshold: every keypoint would be NaN)
    gin = layer.graph_input((1, 1, 256, 256))
    gin.copy_(img[:1])
    eager_out = layer.predict(img[:1])
    g_out = layer.predict_graphed(gin)
    assert torch.isfinite(eager_out.pred_keypoints).all()
    assert torch.equal(g_out.pred_keypoints, eager_out.pred_keypoints) and torch.equal(g_out.pred_peak_values, eager_out.pred_peak_values)
    gin.copy_(img[1:2])
    g2 = layer.predict_graphed(gin)
    e2 = layer.predict(img[1:2])
    assert torch.equal(g2.pred_keypoints, e2.pred_keypoints) and not torch.equal(e2.pred_keypoints, eager_out.pred_keypoints)
    assert torch.equal(layer.predict_graphed(img[:1]).pred_keypoints, eager_out.pred_keypoints)  # a foreign tensor is copied into the graph's buffer
    # the graph's own buffer takes the fast path (replay only): it must notice a replaced post-process config and a refilled buffer all the same
    gin.copy_(img[:1])
    assert torch.equal(layer.predict_graphed(gin).pred_keypoints, eager_out.pred_keypoints)
    layer.postprocess_config = PostprocessConfig(peak_threshold=10.0)  # nothing passes
    assert torch.isnan(layer.predict_graphed(gin).pred_keypoints).all() and torch.isnan(layer.predict(img[:1]).pred_keypoints).all()
    layer.postprocess_config = PostprocessConfig(peak_threshold=-1.0)
    assert torch.equal(layer.predict_graphed(gin).pred_keypoints, eager_out.pred_keypoints)
    # float frames take normalize_on_gpu's data-dependent branch in the graphed step exactly as in predict: 0..255 floats are divided by 255, 0..1 floats are not
    f255 = img[:1].to(torch.float32)
    f01 = f255 / 255.0
    for fin in (f255, f01):
        assert torch.equal(layer.predict_graphed(fin).pred_keypoints, layer.predict(fin).pred_keypoints)
    assert torch.equal(layer.predict_graphed(f255).pred_keypoints, eager_out.pred_keypoints)
    assert (layer.predict_graphed(f01).pred_keypoints - eager_out.pred_keypoints).abs().max().item() <= 0.05  # (x / 255 on the host first: rounding-level differences of the maps)
    # frames of another original size that pad to the same shape share the graph but get their own preprocessing record
    small = layer.predict_graphed(img[:1, :, :250, :250])
    assert small.preprocess_info.original_size == (250, 250) and layer.predict_graphed(img[:1]).preprocess_info.original_size == (256, 256)
    raw = layer.backend(img[:1])["SingleInstanceConfmapsHead"]
    codes = m.last_kernels()
    assert codes.count(L.KV_SMALLMAP) >= 10 and L.KV_WINO2D_KS not in codes, codes  # the default routing of this batch: the stride >= 4 levels on conv3x3_sm_kernel, one launch per layer
    layer.postprocess_config = PostprocessConfig()
    out = layer.postprocess({"SingleInstanceConfmapsHead": raw}, PreprocInfo(eff_scale=torch.ones(1), output_stride=2))
    rk, rv = O.single_instance_postprocess(first, 2)
    assert np.allclose(out.pred_keypoints.cpu().numpy(), rk.numpy(), atol=1e-4, equal_nan=True)
    assert np.array_equal(out.pred_peak_values.cpu().numpy(), rv.numpy())
    raw3 = layer.backend(img)["SingleInstanceConfmapsHead"]  # other routing (more tiles): same frame within the relative bar, all three vs the oracle
    _close(raw3[:1], ref, "cfg1 in a batch of 3")
    m.set_option("conv_smallmap", 0)
    split = m(img[:1].to(DEV))["SingleInstanceConfmapsHead"].cpu()
    assert L.KV_WINO2D_KS in m.last_kernels() and L.KV_SMALLMAP not in m.last_kernels()  # without the small-map kernel: K split over workgroups + a second stage
    _close(split, ref, "cfg1, split-K kernels")
    m.set_option("conv_splitk", 0)
    one_stage = m(img[:1].to(DEV))["SingleInstanceConfmapsHead"].cpu()
    assert L.KV_WINO2D_KS not in m.last_kernels() and L.KV_SMALLMAP not in m.last_kernels()
    _close(one_stage, ref, "cfg1, one-stage kernels")
    assert (one_stage - first).abs().max().item() <= 2e-5 * ref.abs().max().item() and (split - first).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("filters,max_stride,hw,batch,out_stride,splitk", [(32, 8, (32, 32), 1, None, 1), (32, 8, (32, 32), 1, None, 3), (32, 16, (64, 48), 2, 2, 1), (64, 8, (40, 24), 1, None, 5),
                                                                           (32, 8, (40, 56), 1, 4, 2), (32, 16, (48, 80), 1, 2, 7), (16, 32, (64, 64), 1, 4, 1)])
def test_split_k_winograd_kernel_matches_the_one_stage_kernel_and_the_oracle(filters, max_stride, hw, batch, out_stride, splitk):
    """conv3x3_wino2d_kernel<.., KS> + splitk_reduce_kernel: K slices that start / end inside either concat source (decoder convs), one-half slices
    (forced counts up to the number of halves), the fused pool with odd sizes and image-cut tiles through the second stage, auto routing (1) and
    forced slice counts; against the oracle, against the one-stage kernel (conv_splitk = 0) to a few ulp of the tensor's scale, intermediate
    activations included, and bitwise repeatable."""
    from sleap_nn_amd.architectures.model import Model

    os_ = out_stride or max_stride
    bb = dict(SI_BB, filters=filters, max_stride=max_stride, output_stride=os_)
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": os_}}
    sd = O.init_state(bb, heads, "single_instance", seed=filters + hw[0] + splitk, head_scale=1.0)
    g = torch.Generator().manual_seed(hw[1])
    img = torch.randint(0, 256, (batch, 1, hw[0], hw[1]), dtype=torch.uint8, generator=g)
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    outs, kinds = {}, {}
    for name, ks, finish in (("split", splitk, 0), ("in_kernel", splitk, 1), ("shared", splitk, 2), ("one", 0, 0)):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        m.set_option("conv_splitk", ks)
        m.set_option("conv_smallmap", 0)  # (the small-map kernel would take these layers from the split form under the automatic routing)
        m.set_option("conv_splitk_finish", finish)  # 0: splitk_reduce_kernel; 1: the workgroup that stores a unit's last K slice runs the second stage; 2: the unit's workgroups share it
        outs[name] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        kinds[name] = m.last_kernels()
        if name != "one":
            for _ in range(3):  # (the counters of the in-kernel second stage are back at zero after every launch)
                again = m(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
                assert torch.equal(again, outs[name])
    assert L.KV_WINO2D_KS in kinds["split"] and L.KV_WINO2D_KS not in kinds["one"], kinds
    assert torch.equal(outs["split"], outs["in_kernel"]) and torch.equal(outs["split"], outs["shared"])  # the same sums in the same slice order, whichever workgroup adds them
    _close(outs["split"], ref, "split")
    assert (outs["split"] - outs["one"]).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("filters,max_stride,hw,batch,out_stride,reuse", [(16, 16, (64, 64), 1, 2, 1), (16, 16, (80, 96), 2, 2, 1), (32, 8, (40, 24), 3, 2, 1), (16, 32, (96, 160), 1, 4, 1),
                                                                          (8, 16, (48, 80), 2, 1, 1), (16, 16, (48, 112), 2, 2, 1), (24, 8, (40, 56), 1, 4, 1), (16, 16, (80, 96), 2, 2, 0), (32, 16, (32, 48), 1, 16, 1)])
def test_small_map_kernel_on_every_3x3_conv_matches_the_oracle(filters, max_stride, hw, batch, out_stride, reuse):
    """conv3x3_sm_kernel forced onto every 3x3 conv whose shape it takes (conv_smallmap = 2): one and two sources, the concat boundary inside a 32-channel chunk (16 + 32,
    24 -> 32-padded + 64 channels), the bilinear x2 folded into the loader (inference plans) and as a tensor (workspace_reuse = 0), fused pool, an
    unread full-resolution output, maps that cut the 8 x 8-pixel units, three frames; against the oracle, against the kernels that run otherwise, bitwise repeatable."""
    from sleap_nn_amd.architectures.model import Model

    bb = dict(SI_BB, filters=filters, max_stride=max_stride, output_stride=out_stride)
    heads = {"confmaps": {"part_names": [f"n{i}" for i in range(3 + (hw[1] // 8) % 17)], "output_stride": out_stride}}  # (3 .. 19 head channels: fused behind a conv of <= 32 channels)
    sd = O.init_state(bb, heads, "single_instance", seed=filters + hw[0], head_scale=1.0)
    img = torch.randint(0, 256, (batch, 1, hw[0], hw[1]), dtype=torch.uint8, generator=torch.Generator().manual_seed(hw[1]))
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    outs, kinds = {}, {}
    for name, mode in (("sm", 2), ("other", 0)):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        m.set_option("conv_smallmap", mode)
        m.set_option("workspace_reuse", reuse)
        outs[name] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        kinds[name] = m.last_kernels()
        assert torch.equal(m(img.to(DEV))["SingleInstanceConfmapsHead"].cpu(), outs[name])
    n_conv = sum(1 for c in kinds["other"] if c in (L.KV_WINO2D, L.KV_WINO2D_KS, L.KV_W16, L.KV_WINO4, L.KV_C16, L.KV_WINO1D, L.KV_DIRECT, L.KV_ROWGEMM))
    assert kinds["sm"].count(L.KV_SMALLMAP) == n_conv and L.KV_SMALLMAP not in kinds["other"], (kinds, n_conv)
    fused = L.KV_FUSED in kinds["sm"]
    assert fused == (filters * 2 ** {1: 0, 2: 1, 4: 2, 8: 3, 16: 4}[out_stride] <= 32), kinds["sm"]  # the head rides in the last conv's epilogue when that conv has <= 32 channels
    _close(outs["sm"], ref, "small-map kernel")
    assert (outs["sm"] - outs["other"]).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_split_k_second_stage_pool_padding_and_unread_full_resolution_output():
    """The second stage writes the fused 2x2 max pool with "same" zero padding on odd sizes (common.py:69-107) and honours skip_dst (an
    inference plan whose full-resolution conv output nobody reads): a deep encoder on a 34 x 38 map, forced split, vs the oracle's activations."""
    from sleap_nn_amd.architectures.model import Model

    bb = dict(SI_BB, filters=32, max_stride=8, output_stride=8)
    heads = {"confmaps": {"part_names": ["a", "b"], "output_stride": 8}}
    sd = O.init_state(bb, heads, "single_instance", seed=3, head_scale=1.0)
    img = torch.randint(0, 256, (2, 1, 136, 152), dtype=torch.uint8, generator=torch.Generator().manual_seed(8))
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    for reuse in (0, 1):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        m.set_option("conv_splitk", 4)
        m.set_option("workspace_reuse", reuse)
        got = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        assert L.KV_WINO2D_KS in m.last_kernels()
        _close(got, ref, f"reuse={reuse}")


def test_published_workload_fixture_bottomup_batch4_320x560():
    """The one workload the reference publishes numbers for (docs/guides/inference-performance.md:40-48,70-77): its fixture bottom-up run directory
    (UNet f16 / rate 1.5 / max_stride 8, transposed-conv decoder, 2 nodes / 1 edge) at 320 x 560, batch 4 (predictor.py:884,930).  The fixture's weights are
    pinned against the reference itself by ckpt_bottomup.npz at 384 x 384; here the same run directory at the published frame size and batch: network vs
    the oracle, then peaks / grouping of the whole layer vs the oracle's post-process on the oracle's maps."""
    from sleap_nn_amd.inference.backends import HipBackend
    from sleap_nn_amd.inference.layers import BottomUpLayer, PostprocessConfig
    from sleap_nn_amd.inference.loaders import load_model_assets
    from sleap_nn_amd.inference.ops.paf import PAFScorer

    a = load_model_assets(os.path.join(G.GOLDEN_DIR, "ckpt_dirs", "minimal_instance_bottomup"))
    m = a.build_model()
    z = G.load("ckpt_bottomup.npz")
    frames = torch.from_numpy(z["image"]).squeeze(1)  # (2, 1, 384, 384) real frames of the fixture video
    img = torch.cat([frames, frames.flip(-1)], 0)[:, :, 32:352, :].repeat(1, 1, 1, 2)[..., :560].contiguous()  # (4, 1, 320, 560) of real texture
    assert tuple(img.shape) == (4, 1, 320, 560) and img.dtype == torch.uint8
    sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in a.state_dict.items()}  # LightningModule keys (loaders.py:144-176)
    ref = O.model_forward(sd, a.backbone_config, a.head_config, "bottomup", img)
    cs, ps = a.head_config["confmaps"]["output_stride"], a.head_config["pafs"]["output_stride"]
    for use_graph in (False, True):
        layer = BottomUpLayer(HipBackend(m, DEV, use_graph=use_graph), PAFScorer.from_config(a.head_config), cs, ps, max_stride=a.backbone_config["max_stride"],
                              postprocess_config=PostprocessConfig(peak_threshold=0.2))
        raw = layer.backend(img)
        for k, v in ref.items():
            err = (raw[k].cpu() - v).abs().max().item()
            assert err <= CMS_ATOL, (k, err)
        out = layer.predict(img)
        rk, rv, rs = O.bottomup_postprocess(ref["MultiInstanceConfmapsHead"], ref["PartAffinityFieldsHead"],
                                            O.PAFScorerRef(a.head_config["confmaps"]["part_names"], [tuple(e) for e in a.head_config["pafs"]["edges"]], ps), cs, peak_threshold=0.2)
        k = out.pred_keypoints.numpy()
        assert k.shape == rk.shape and np.array_equal(np.isnan(k), np.isnan(rk))
        assert np.allclose(k, rk, atol=1e-3, equal_nan=True)
        assert np.isfinite(k).any()  # real detections, not an all-NaN agreement


def test_pipelined_predictor_graphed_gpu_stage_equals_the_eager_layer():
    """Predictor.predict on the published workload's run directory (batch 4, 320 x 560): the pipelined path -- pinned staging, forward + peaks + candidate scoring as ONE
    hipGraph per shape, one D2H, ph_group_packed in the worker -- returns bit for bit what the layer's eager predict returns batch by batch; a ragged last batch (another
    graph), frames already on the device, max_instances (truncation by score) and the capacity-overflow redo (capacities forced tiny) included; and the eager
    pipelined form (use_graph=False) agrees too."""
    from sleap_nn_amd.inference.layers import PostprocessConfig
    from sleap_nn_amd.inference.predictor import Predictor

    root = os.path.join(G.GOLDEN_DIR, "ckpt_dirs", "minimal_instance_bottomup")
    z = G.load("ckpt_bottomup.npz")
    two = torch.from_numpy(z["image"]).squeeze(1)
    vid = torch.cat([two, two.flip(-1)], 0)[:, :, 32:352, :].repeat(4, 1, 1, 2)[..., :560].contiguous()[:14]  # 14 frames: batches of 4, 4, 4, 2
    pred = Predictor.from_model_paths([root], device=DEV, batch_size=4, peak_threshold=0.2)
    ref = pred.predict(vid, pipelined=False)
    assert len(ref) == 4 and sum(int(torch.isfinite(o.instance_scores).sum()) for o in ref) >= 14

    def same(outs, refs):
        assert len(outs) == len(refs)
        for o, r in zip(outs, refs):
            assert torch.equal(o.frame_indices, r.frame_indices)
            for f in ("pred_keypoints", "pred_peak_values", "instance_scores"):
                a, b = getattr(o, f).numpy(), getattr(r, f).numpy()
                assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), f

    for _ in range(2):  # (second pass: replays only)
        same(pred.predict(vid), ref)
    same(pred.predict(vid.to(DEV)), ref)
    same(pred.predict(vid.numpy()), ref)
    pred.use_graph = False
    same(pred.predict(vid), ref)
    pred.use_graph = True
    # frames without a single peak (all-NaN outputs of one instance slot), an empty list of frames, a single frame
    blank = torch.zeros_like(vid[:6])
    refb = pred.predict(blank, pipelined=False)
    assert all(torch.isnan(o.pred_keypoints).all() and o.pred_keypoints.shape[1] == 1 for o in refb)
    same(pred.predict(blank), refb)
    assert pred.predict(vid[:0]) == []
    same(pred.predict(vid[:1]), pred.predict(vid[:1], pipelined=False))
    pred.layer.postprocess_config = PostprocessConfig(peak_threshold=0.2, max_instances=1)
    ref1 = pred.predict(vid, pipelined=False)
    assert ref1[0].pred_keypoints.shape[1] == 1
    same(pred.predict(vid), ref1)
    # a caller that wants the maps back: the worker redoes such a batch eagerly on the stream it was enqueued on (three copies of the layer on three streams here)
    pc = Predictor.from_model_paths([root], device=DEV, batch_size=4, peak_threshold=0.2, return_confmaps=True)
    assert len(pc.replicas) == 2  # (from_model_paths(streams=3): three copies of a small network)
    refc = pc.predict(vid, pipelined=False)
    gotc = pc.predict(vid)
    same(gotc, refc)
    for o, r in zip(gotc, refc):
        assert o.pred_confmaps is not None and torch.equal(o.pred_confmaps, r.pred_confmaps.cpu())
    # capacity overflow: a fresh layer whose captured capacities are too small for the frames redoes the batch eagerly with larger ones -- same results
    pred2 = Predictor.from_model_paths([root], device=DEV, batch_size=4, peak_threshold=0.2)
    pred2.layer._capacities = lambda B, n, _l=pred2.layer: (max(_l._peak_cap, 2), max(_l._cand_cap, 1))
    same(pred2.predict(vid), ref)
    assert pred2.layer._peak_cap > 2


def test_centroid_selection_kernel_topk_padding_and_lists():
    """ph_centroid_select against the torch statement of CentroidLayer.postprocess / TopDownLayer's list building: frames with fewer peaks than max_instances (kept in order), with
    more (torch.topk order, descending), with none; NaN padding, input-scale and eff_scale undo, boxes (make_centered_bboxes), stage-2 lists in nonzero order."""
    import ctypes as C

    from sleap_nn_amd.inference.ops.crops import make_centered_bboxes

    rng = np.random.RandomState(0)
    B, I, cap = 5, 4, 64
    per = [0, 3, 9, 4, 1]
    n = sum(per)
    xy = torch.from_numpy(rng.rand(cap, 2).astype(np.float32) * 300)
    vals = torch.from_numpy(rng.rand(cap).astype(np.float32))
    offs = np.concatenate([[0], np.cumsum(per)]).astype(np.int32)
    counts = torch.from_numpy(np.concatenate([[n], per, offs]).astype(np.int32))
    eff = torch.tensor([1.0, 0.5, 2.0, 1.25, 1.0])
    input_scale, ch, cw = 0.5, 48, 64
    d = lambda t: t.to(DEV)
    cp, cv = torch.empty((B, I, 2), device=DEV), torch.empty((B, I), device=DEV)
    bb = torch.empty((B, I, 4, 2), device=DEV)
    ls, lt, lslot, pos = (torch.full((B * I,), -7, dtype=torch.int32, device=DEV), torch.zeros((B * I, 2), device=DEV), torch.full((B * I,), -7, dtype=torch.int32, device=DEV),
                          torch.full((B * I,), -7, dtype=torch.int32, device=DEV))
    nv = torch.zeros(1, dtype=torch.int32, device=DEV)
    xd, vd, cd, ed = d(xy), d(vals), d(counts), d(eff)
    P = lambda t: C.c_void_p(t.data_ptr())
    L.check(L.lib().ph_centroid_select(P(xd), P(vd), P(cd), B, I, cap, input_scale, P(ed), float(ch), float(cw), P(cp), P(cv), P(bb), P(ls), P(lt), P(lslot), P(pos), P(nv), None))
    torch.cuda.synchronize()
    ref_cp, ref_cv = torch.full((B, I, 2), float("nan")), torch.full((B, I), float("nan"))
    for b in range(B):
        p_, v_ = xy[offs[b] : offs[b + 1]] / input_scale, vals[offs[b] : offs[b + 1]]
        if per[b] > I:
            v_, idx = torch.topk(v_, I)
            p_ = p_[idx]
        k = min(per[b], I)
        ref_cp[b, :k], ref_cv[b, :k] = p_[:k] / eff[b], v_[:k]
    assert np.array_equal(cp.cpu().numpy(), ref_cp.numpy(), equal_nan=True) and np.array_equal(cv.cpu().numpy(), ref_cv.numpy(), equal_nan=True)
    valid = ~torch.isnan(ref_cp).any(-1)
    idx = valid.nonzero()
    nvalid = int(idx.shape[0])
    assert int(nv.item()) == nvalid == sum(min(c, I) for c in per)
    # stage 2 works in SIZED space (reference layers/topdown.py:127-150, 262-267): boxes around centroid * eff_scale, top-left list in sized space (the crops are cut
    # from the sizematched frame), the stored boxes / eff_scale; eff_scale here is 0.5 / 2 / 1.25 on three of the five frames (the round-5 kernel boxed the
    # image-space centroid, i.e. cut stage-2 crops at another scale than the reference whenever the sizematcher was active)
    per_eff = eff[idx[:, 0]].view(-1, 1, 1)
    ref_bb_sized = make_centered_bboxes(ref_cp[idx[:, 0], idx[:, 1]] * per_eff.view(-1, 1), ch, cw)
    ref_bb = ref_bb_sized / per_eff
    assert np.array_equal(bb.cpu()[idx[:, 0], idx[:, 1]].numpy(), ref_bb.numpy()) and torch.isnan(bb.cpu()[~valid]).all()
    assert torch.equal(ls.cpu()[:nvalid], idx[:, 0].int()) and torch.equal(lslot.cpu()[:nvalid], (idx[:, 0] * I + idx[:, 1]).int())
    assert np.array_equal(lt.cpu()[:nvalid].numpy(), ref_bb_sized[:, 0].numpy())
    ref_pos = torch.full((B * I,), -1, dtype=torch.int32)
    ref_pos[(idx[:, 0] * I + idx[:, 1])] = torch.arange(nvalid, dtype=torch.int32)
    assert torch.equal(pos.cpu(), ref_pos)
    # ... and the way back
    N = 3
    k3, v3 = torch.from_numpy(rng.rand(nvalid, N, 2).astype(np.float32) * 40), torch.from_numpy(rng.rand(nvalid, N).astype(np.float32))
    fk, fc, fv = torch.empty((B * I, N, 2), device=DEV), torch.empty((B * I, N, 2), device=DEV), torch.empty((B * I, N), device=DEV)
    k3d, v3d = d(k3), d(v3)
    L.check(L.lib().ph_topdown_scatter(P(k3d), P(v3d), P(lt), P(pos), B * I, N, P(ed), I, P(fk), P(fc), P(fv), None))
    torch.cuda.synchronize()
    rk, rc, rv = torch.full((B * I, N, 2), float("nan")), torch.full((B * I, N, 2), float("nan")), torch.full((B * I, N), float("nan"))
    flat = idx[:, 0] * I + idx[:, 1]
    rk[flat], rc[flat], rv[flat] = (k3 + ref_bb_sized[:, 0].view(-1, 1, 2)) / per_eff, k3, v3  # add_crop_offset in sized space, then / eff_scale
    for got, want in ((fk, rk), (fc, rc), (fv, rv)):
        assert np.array_equal(got.cpu().numpy(), want.numpy(), equal_nan=True)


@pytest.mark.parametrize("filters,max_stride,hw,batch,out_stride", [(16, 4, (64, 96), 2, 1), (16, 8, (72, 40), 1, 1), (8, 8, (64, 48), 1, 2), (8, 4, (64, 80), 3, 2)])
def test_wave_private_kernel_takes_the_two_source_decoder_conv(filters, max_stride, hw, batch, out_stride):
    """conv3x3_w16_kernel<CHUNKS, 1, TWO>: the full-resolution decoder level of an output-stride-1 filters-16 UNet -- concat(skip 16, up-sampled 32) -> 16 channels --
    and the last level of a filters-8 one (16 + 32 -> 16 at half resolution) as two K panels with their own descriptors; image-cut tiles, several tiles per workgroup;
    vs the oracle and vs the F(2,3) kernel the layer ran on before (conv_w16 = 0)."""
    from sleap_nn_amd.architectures.model import Model

    bb = dict(SI_BB, filters=filters, max_stride=max_stride, output_stride=out_stride)
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": out_stride}}
    sd = O.init_state(bb, heads, "single_instance", seed=filters + hw[0], head_scale=1.0)
    img = torch.randint(0, 256, (batch, 1, hw[0], hw[1]), dtype=torch.uint8, generator=torch.Generator().manual_seed(hw[1]))
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    outs = {}
    for name, w16 in (("w16", 1), ("w1d", 0)):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        m.set_option("conv_w16", w16)
        m.set_option("conv_smallmap", 0)  # (the small-map kernel would take the layer at these sizes)
        outs[name] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        codes, tab = m.last_kernels(), m.op_table(batch, hw[0], hw[1])
        last_concat = [c for r, c in zip(tab, codes) if "refine_conv0" in r["label"]][-1]
        assert (last_concat == L.KV_W16) == bool(w16), (name, last_concat)
    _close(outs["w16"], ref, "two-source w16")
    assert (outs["w16"] - outs["w1d"]).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("filters,out_stride,hw,batch,nodes", [(16, 2, (64, 96), 2, 5), (16, 2, (72, 128), 1, 13), (16, 1, (32, 64), 2, 3), (8, 2, (80, 64), 1, 16), (16, 2, (64, 64), 1, 17)])
def test_head_fused_into_the_wave_private_kernel_matches_the_head_kernel_and_the_oracle(filters, out_stride, hw, batch, nodes):
    """conv3x3_w16_kernel<.., HEAD>: the 1x1 head behind the final 32 -> 32 (output stride 2) or 16 -> 16 (output stride 1, filters 8) conv as MFMAs on the epilogue's registers;
    image-cut tiles, 16 head channels (the limit) and 17 (not fused: the head kernel runs); head_fuse = 0 is the separate launch."""
    from sleap_nn_amd.architectures.model import Model

    bb = dict(SI_BB, filters=filters, max_stride=4, output_stride=out_stride)
    heads = {"confmaps": {"part_names": [f"n{i}" for i in range(nodes)], "output_stride": out_stride}}
    sd = O.init_state(bb, heads, "single_instance", seed=nodes, head_scale=1.0)
    img = torch.randint(0, 256, (batch, 1, hw[0], hw[1]), dtype=torch.uint8, generator=torch.Generator().manual_seed(hw[0]))
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    outs = {}
    for fuse in (1, 0):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        m.set_option("head_fuse", fuse)
        m.set_option("conv_smallmap", 0)  # (the small-map kernel, which would run these convs and carries heads of any width, has its own test)
        m.eval()
        outs[fuse] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        head_codes = [c for r, c in zip(m.op_table(batch, hw[0], hw[1]), m.last_kernels()) if r["kind"] == L.OP_HEAD]
        assert head_codes == [L.KV_FUSED if (fuse and nodes <= 16) else L.KV_NONE], (fuse, head_codes)  # a fused head's op launches nothing
    _close(outs[1], ref, "fused head")
    assert (outs[1] - outs[0]).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("filters,max_stride,hw,batch", [(16, 16, (128, 160), 2), (16, 4, (72, 104), 1), (16, 8, (64, 64), 3)])
def test_cout32_layers_on_the_half_empty_n_tile_of_the_winograd_kernel(filters, max_stride, hw, batch):
    """conv_n32_wino2d: the last decoder level of an output-stride-2 filters-16 UNet (concat 32 + 64 -> 32 channels) runs on conv3x3_wino2d_kernel<64, HT> with the upper half
    of its N tile empty (a second weight packing, zero rows 32 .. 63; stores skip the missing channels) instead of the N-tile-32 F(2,3) kernel; image-cut tiles; vs the oracle
    and vs the old route (option 0), and the kernel record says which ran."""
    from sleap_nn_amd.architectures.model import Model

    bb = dict(SI_BB, filters=filters, max_stride=max_stride, output_stride=2)
    heads = {"confmaps": {"part_names": ["a", "b", "c", "d"], "output_stride": 2}}
    sd = O.init_state(bb, heads, "single_instance", seed=hw[0], head_scale=1.0)
    img = torch.randint(0, 256, (batch, 1, hw[0], hw[1]), dtype=torch.uint8, generator=torch.Generator().manual_seed(hw[1]))
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    outs = {}
    for opt in (1, 0):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        m.set_option("conv_n32_wino2d", opt)
        m.set_option("conv_smallmap", 0)  # (at these batch sizes the small-map kernel would take the layer under test)
        outs[opt] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        last_concat = [c for r, c in zip(m.op_table(batch, hw[0], hw[1]), m.last_kernels()) if "refine_conv0" in r["label"]][-1]
        assert last_concat == (L.KV_WINO2D if opt else L.KV_WINO1D), (opt, last_concat)
    _close(outs[1], ref, "n32 on wino2d")
    assert (outs[1] - outs[0]).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # the same layer on the F(4x4,3x3) kernel (its N-tile-64 packing again, the waves of the empty N half skip their MFMAs, the bilinear x2 of its second source folded in):
    # what cfg2's 96 -> 32 layer at 256 x 256 x 8 takes by the time model; forced here
    m = Model("unet", bb, heads, "single_instance")
    m.load_state_dict(sd)
    m.set_option("conv_wino4", 3)
    m.set_option("conv_smallmap", 0)
    out4 = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
    last_concat = [c for r, c in zip(m.op_table(batch, hw[0], hw[1]), m.last_kernels()) if "refine_conv0" in r["label"]][-1]
    assert last_concat == L.KV_WINO4, last_concat
    _close(out4, ref, "n32 on wino4")


@pytest.mark.parametrize("filters,max_stride,hw,batch,out_stride,splitk", [(32, 8, (64, 64), 1, 4, 3), (32, 16, (64, 96), 2, 4, 2), (16, 32, (128, 128), 1, 4, 5), (16, 32, (256, 256), 2, 4, 1)])
def test_split_k_form_of_the_f4x4_kernel_matches_the_one_stage_kernel_and_the_oracle(filters, max_stride, hw, batch, out_stride, splitk):
    """conv3x3_wino4_kernel<.., KS> + splitk_reduce_kernel: K slices in quarters through both concat sources incl. the folded-bilinear (half-resolution) second source, forced slice
    counts (conv_wino4 = 3 puts every fitting layer on the kernel) and the default cost-routed choice (1: F(4x4,3x3) vs F(2x2,3x3), each priced at the split it would take); vs the oracle
    at the F(4x4,3x3) bar (1e-5 of the tensor's scale), vs the unsplit kernels, bitwise repeatable."""
    from sleap_nn_amd.architectures.model import Model

    bb = dict(SI_BB, filters=filters, max_stride=max_stride, output_stride=out_stride)
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "output_stride": out_stride}}
    sd = O.init_state(bb, heads, "single_instance", seed=filters + hw[0] + splitk, head_scale=1.0)
    img = torch.randint(0, 256, (batch, 1, hw[0], hw[1]), dtype=torch.uint8, generator=torch.Generator().manual_seed(hw[1] + splitk))
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    outs = {}
    for name, ks in (("split", splitk), ("one", 0)):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        if splitk != 1:
            m.set_option("conv_wino4", 3)
        m.set_option("conv_splitk", ks)
        m.set_option("conv_smallmap", 0)  # (this test is about the split forms; the small-map kernel has its own)
        outs[name] = m.to(DEV)(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
        assert torch.equal(m(img.to(DEV))["SingleInstanceConfmapsHead"].cpu(), outs[name])
        if name == "split" and splitk != 1:
            assert L.KV_WINO4 in m.last_kernels()
    _close(outs["split"], ref, "wino4 split")
    assert (outs["split"] - outs["one"]).abs().max().item() <= 3e-5 * ref.abs().max().item()


def test_lane_streams_are_chosen_by_a_measured_overlap():
    """predictor.concurrent_streams: the lanes of a two-stream predictor must not share a hardware queue whatever number of streams the process created before (the runtime deals
    streams over a handful of queues): a short launch on one lane finishes while the other spins, for several counts of earlier streams."""
    from sleap_nn_amd.inference.predictor import concurrent_streams

    dev = torch.device(DEV)
    keep = []
    for before in (0, 1, 2, 3):
        keep += [torch.cuda.Stream(dev) for _ in range(before)]
        for st in keep:
            with torch.cuda.stream(st):
                torch.zeros(8, device=dev).add_(1)
        a, b = concurrent_streams(dev, 2)
        assert a != b
        x = torch.zeros(64, device=dev)
        torch.cuda.synchronize()
        for first, second in ((a, b), (b, a)):
            busy, done = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.stream(first):
                torch.cuda._sleep(4_000_000)
                busy.record(first)
            with torch.cuda.stream(second):
                x.add_(1.0)
                done.record(second)
            done.synchronize()
            assert not busy.query(), f"lane streams run in order ({before} streams made before)"
            busy.synchronize()
