"""GPU parity of the ConvNeXt encoder-decoder path vs the oracle.

The oracle's CNBlock / LayerNorm2d arithmetic restates torchvision's public definition (torchvision
is absent from this image: parity of that half is self-consistent, see oracle/cpu_ref.py:convnext_plan);
its pool / middle / decoder half is pinned against the reference's own modules
(tests/test_oracle_golden.py::test_convnext_wrapper_decoder_half_matches_reference).
Tolerance: 1e-4 absolute on head outputs and intermediate activations (north_star).
"""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ATOL = 1e-4


def _bb(**kw):
    bb = {"model_type": None, "arch": None, "in_channels": 1, "kernel_size": 3, "filters_rate": 2, "convs_per_block": 2, "up_interpolate": True,
          "stem_patch_kernel": 4, "stem_patch_stride": 2, "output_stride": 2, "max_stride": 32}
    bb.update(kw)
    return bb


def _heads(n, stride):
    return {"confmaps": {"part_names": [str(i) for i in range(n)], "sigma": 2.5, "output_stride": stride}}


def _run(bb, heads, model_type, img, seed=7, layer_scale=0.5, check_blocks=True):
    from sleap_nn_amd.architectures.model import Model

    sd = O.init_state_convnext(bb, heads, model_type, seed=seed, head_scale=1.0, layer_scale=layer_scale, randomize_affine=True)
    m = Model("convnext", bb, heads, model_type)
    m.load_state_dict(sd, strict=True)
    m.to(DEV).set_keep_activations(check_blocks)
    collect = {}
    ref = O.model_forward(sd, bb, heads, model_type, img, collect=collect, backbone="convnext")
    out = m(img.to(DEV))
    torch.cuda.synchronize()
    worst = {}
    if check_blocks:
        for name, t in collect.items():
            if name not in m.backbone.labels:
                continue
            got = m.read_activation(name, t.shape[0], t.shape[-2:]).cpu()
            scale = max(1.0, t.abs().max().item())
            err = (got - t).abs().max().item() / scale
            worst[name] = err
            assert err <= ATOL, (name, err, scale)
    for k, t in ref.items():
        got = out[k].cpu()
        assert got.shape == t.shape
        err = (got - t).abs().max().item() / max(1.0, t.abs().max().item())
        assert err <= ATOL, (k, err)
    return worst


@pytest.mark.parametrize(
    "channels,depths,stem_stride,os_,hw,batch",
    [
        ([16, 32, 64, 128], [1, 2, 1, 1], 2, 2, (64, 96), 2),      # multiples of 16
        ([24, 40, 72, 136], [2, 1, 1, 1], 2, 4, (64, 64), 3),      # padded channels: LayerNorm must ignore the pad lanes
        ([32, 64, 128, 256], [1, 1, 2, 1], 4, 1, (128, 64), 1),    # stem stride 4: two decoder blocks without a skip
        ([96, 192, 384, 768], [1, 1, 1, 1], 2, 2, (96, 160), 2),   # the tiny variant's widths (BN = 96 / 128 GEMM tiles)
    ],
)
def test_convnext_forward_matches_oracle(channels, depths, stem_stride, os_, hw, batch):
    bb = _bb(arch={"depths": depths, "channels": channels}, stem_patch_stride=stem_stride, output_stride=os_)
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (batch, 1, hw[0], hw[1]), dtype=torch.uint8, generator=g)
    _run(bb, _heads(5, os_), "single_instance", img)


def test_convnext_tiny_centered_instance_rgb_float():
    """Full ConvNeXt-tiny (87.9 M parameters: exercises the > 2^24-parameter gather maps) on a float RGB crop."""
    bb = _bb(model_type="tiny", in_channels=3, output_stride=2)
    g = torch.Generator().manual_seed(3)
    img = torch.rand((1, 3, 96, 96), generator=g)
    _run(bb, _heads(13, 2), "centered_instance", img, layer_scale=0.3, check_blocks=False)


def test_cnblock_mlp_in_one_launch_equals_the_two_row_gemms_and_the_oracle():
    """Inference plans run a 96- or 192-channel CNBlock's Linear -> GELU -> Linear -> layer scale + residual as ONE launch (cnblock_mlp_kernel: the second product reads the first one's
    accumulator registers as its B operand, K orders permuted to match): against the oracle at the path's tolerance, against the two-GEMM plan (`mlp_fuse` 0) at fp32
    summation-order noise, over several tiles and frames; both Linear ops of a fused pair report
    PH_KV_MLP, the plan that keeps activations (hidden tensor readable) does not fuse."""
    from sleap_nn_amd import _lib as L
    from sleap_nn_amd.architectures.model import Model

    bb = _bb(arch={"depths": [2, 1, 1, 1], "channels": [96, 192, 384, 768]}, stem_patch_stride=2, output_stride=2)
    heads = _heads(4, 2)
    g = torch.Generator().manual_seed(23)
    img = torch.randint(0, 256, (3, 1, 64, 96), dtype=torch.uint8, generator=g)  # stage 0: 3 x 32 x 48 pixels = 18 tiles of 256 rows (a valid input's pixel count is always a multiple of 256 at this stage)
    sd = O.init_state_convnext(bb, heads, "single_instance", seed=5, head_scale=1.0, layer_scale=0.5, randomize_affine=True)
    ref = O.model_forward(sd, bb, heads, "single_instance", img, backbone="convnext")
    outs = {}
    for fuse in (1, 0):
        m = Model("convnext", bb, heads, "single_instance")
        m.load_state_dict(sd, strict=True)
        m.to(DEV).set_option("mlp_fuse", fuse)
        outs[fuse] = {k: v.clone() for k, v in m(img.to(DEV)).items()}
        codes = list(m.last_kernels())
        assert codes.count(L.KV_MLP) == (6 if fuse else 0), codes  # (two blocks at 96 channels + one at 192) x two Linear ops; the wider stages stay on the row GEMM
    kept = Model("convnext", bb, heads, "single_instance")
    kept.load_state_dict(sd, strict=True)
    kept.to(DEV).set_keep_activations(True)
    kept(img.to(DEV))
    assert list(kept.last_kernels()).count(L.KV_MLP) == 0
    for k, t in ref.items():
        scale = max(1.0, t.abs().max().item())
        assert (outs[1][k].cpu() - t).abs().max().item() / scale <= ATOL, k
        assert (outs[1][k] - outs[0][k]).abs().max().item() / scale <= 2e-5, k
    # the fused kernel's weight images are rebuilt from a device parameter arena too (ph_model_set_params' gather maps: what an evaluation in the middle of training runs on)
    sd2 = O.init_state_convnext(bb, heads, "single_instance", seed=6, head_scale=1.0, layer_scale=0.4, randomize_affine=True)
    ref2 = O.model_forward(sd2, bb, heads, "single_instance", img, backbone="convnext")
    donor = Model("convnext", bb, heads, "single_instance")
    donor.load_state_dict(sd2, strict=True)
    live = Model("convnext", bb, heads, "single_instance")
    live.load_state_dict(sd, strict=True)  # (the handle is created from THESE weights, then re-packed from the arena)
    live.bind_live_params(donor.flat_params().to(DEV))
    out2 = live.to(DEV)(img.to(DEV))
    assert list(live.last_kernels()).count(L.KV_MLP) == 6
    for k, t in ref2.items():
        assert (out2[k].cpu() - t).abs().max().item() / max(1.0, t.abs().max().item()) <= ATOL, k


def test_convnext_rows_not_multiple_of_tile_and_gray_to_rgb():
    """M (pixels) not a multiple of the 256-row GEMM tile at every stage; gray input into an RGB model."""
    bb = _bb(arch={"depths": [1, 1, 1, 1], "channels": [16, 32, 64, 128]}, in_channels=3)
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (1, 1, 32, 96), dtype=torch.uint8, generator=g)
    _run(bb, _heads(3, 2), "single_instance", img)


def test_small_map_convs_row_gemm_equals_halo_kernel():
    """3x3 convolutions on small feature maps run as 9-tap row GEMMs (two-source concat included);
    the halo-tiled kernel must give the same maps (both vs the oracle, and vs each other)."""
    from sleap_nn_amd.architectures.model import Model

    bb = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 16, "output_stride": 2, "convs_per_block": 2,
          "middle_block": True, "up_interpolate": True, "stacks": 1, "stem_stride": None}
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "sigma": 2.5, "output_stride": 2}}
    sd = O.init_state(bb, heads, "single_instance", seed=21, head_scale=1.0)
    g = torch.Generator().manual_seed(8)
    img = torch.randint(0, 256, (3, 1, 80, 48), dtype=torch.uint8, generator=g)  # maps 80x48 ... 5x3
    ref = O.model_forward(sd, bb, heads, "single_instance", img)["SingleInstanceConfmapsHead"]
    outs = {}
    for tag, thr in (("gemm", 2.0), ("halo", 0.0)):
        m = Model("unet", bb, heads, "single_instance")
        m.load_state_dict(sd)
        m.to(DEV).set_option("conv_gemm_fill", thr)  # per-handle option (the library reads no environment)
        outs[tag] = m(img.to(DEV))["SingleInstanceConfmapsHead"].cpu()
    for tag, o in outs.items():
        assert (o - ref).abs().max().item() <= ATOL, tag
    assert (outs["gemm"] - outs["halo"]).abs().max().item() <= 2e-5


# ------------------------------------------------------------------------------------------
# Training: forward + MSE + backward of the ConvNeXt program vs autograd over the oracle.
# Tolerance: loss 1e-5 relative; every parameter gradient within 2e-4 of its tensor's max magnitude.
# ------------------------------------------------------------------------------------------
def _train_case(channels, depths, stem_stride, os_, hw, B, model_type="centered_instance", in_ch=1, seed=13, bb=None, layer_scale=0.6, n_parts=3):
    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.training.module import TrainingModule

    if bb is None:
        bb = _bb(arch={"depths": depths, "channels": channels}, stem_patch_stride=stem_stride, output_stride=os_, in_channels=in_ch)
    heads = {"confmaps": {"part_names": [chr(97 + i) for i in range(n_parts)], "sigma": 2.5, "output_stride": os_, "loss_weight": 1.0, "anchor_part": None}}
    sd = O.init_state_convnext(bb, heads, model_type, seed=seed, head_scale=1.0, layer_scale=layer_scale, randomize_affine=True)
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (B, in_ch, hw[0], hw[1]), dtype=torch.uint8, generator=g)
    ref_out = O.model_forward(sd, bb, heads, model_type, img, backbone="convnext")
    targets = {k: torch.rand(v.shape, generator=g) * 0.5 for k, v in ref_out.items()}
    m = Model("convnext", bb, heads, model_type)
    m.load_state_dict(sd)
    tm = TrainingModule(m, DEV, loss_weights=[1.0])
    losses, ref_grads = O.training_step(sd, bb, heads, model_type, img, targets, [1.0], backbone="convnext")
    loss = tm.forward_backward(img.to(DEV), {k: v.to(DEV) for k, v in targets.items()})
    torch.cuda.synchronize()
    assert abs(float(loss[0]) - losses[0]) <= 1e-5 * max(1.0, abs(losses[0]))
    got = tm.named_grads()
    assert set(got) == set(ref_grads)
    worst = {}
    for k, r in ref_grads.items():
        scale = max(float(r.abs().max()), 1e-12)
        err = float((got[k].cpu() - r).abs().max()) / scale
        worst[k] = err
    bad = {k: v for k, v in worst.items() if v > 2e-4}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    return tm, sd, bb, heads, img, targets


@pytest.mark.parametrize(
    "channels,depths,stem_stride,os_,hw,B",
    [
        ([16, 32, 64, 128], [1, 2, 1, 1], 2, 2, (64, 96), 2),
        ([24, 40, 72, 136], [2, 1, 1, 1], 2, 4, (64, 64), 3),   # padded channels in every reduction
        ([32, 64, 128, 256], [1, 1, 1, 1], 4, 2, (128, 64), 1),  # stem stride 4
    ],
)
def test_convnext_backward_matches_autograd(channels, depths, stem_stride, os_, hw, B):
    _train_case(channels, depths, stem_stride, os_, hw, B)


def test_convnext_adam_steps_reduce_loss_and_match_reference_optimizer():
    tm, sd, bb, heads, img, targets = _train_case([16, 32, 64, 128], [1, 1, 1, 1], 2, 2, (64, 64), 2, seed=29)
    # three full steps against torch.optim.Adam driven by the oracle's autograd gradients
    cur = {k: v.clone() for k, v in sd.items()}
    opt_params = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items()}
    opt = torch.optim.Adam(list(opt_params.values()), lr=1e-3)
    tm.lr = 1e-3
    batch = {"image": img.to(DEV), **{k: v.to(DEV) for k, v in targets.items()}}
    first = None
    tiny = {k: torch.zeros_like(v, dtype=torch.bool) for k, v in sd.items()}
    for step in range(3):
        _, grads = O.training_step({k: p.detach() for k, p in opt_params.items()}, bb, heads, "centered_instance", img, targets, [1.0], backbone="convnext")
        for k, p in opt_params.items():
            p.grad = grads[k]
            tiny[k] |= (grads[k].abs() < 1e-3 * grads[k].abs().max()) & (grads[k] != 0)  # (an exactly zero gradient -- a tap that only sees padding -- moves nothing on either side)
        opt.step()
        loss = tm.training_step(batch)
        first = float(loss[0]) if first is None else first
    torch.cuda.synchronize()
    got = tm.state_dict()
    # An Adam step moves every parameter by at most ~lr whatever the gradient's size, so fp32 noise on a gradient entry shows up at a fixed
    # fraction of lr: 2 % of the total possible movement (3 steps x lr) for every entry whose gradient is not tiny.  An entry whose gradient is
    # below 1e-3 of its tensor's largest in some step is a sum that nearly cancels -- the order of the fp32 additions (the oracle's as much as
    # the kernels': the row weight-gradient GEMM pairs the rows of a chunk differently since round 3) decides a good part of its normalised
    # step m / sqrt(v); those entries (a few per cent of a tensor) get 10 % of the total movement.  Measured: 1.6 % / 3.2 %.
    for k, p in opt_params.items():
        d = (got[k].cpu() - p.detach()).abs()
        big = d[~tiny[k]]
        assert big.numel() == 0 or float(big.max()) <= 0.02 * 3 * 1e-3, k
        assert float(d.max()) <= 0.10 * 3 * 1e-3, k
    assert sum(int(t.sum()) for t in tiny.values()) < 0.5 * sum(t.numel() for t in tiny.values())  # the strict bar covers most of the model
    assert float(loss[0]) < first


# ------------------------------------------------------------------------------------------
# BASELINE cfg4: the ConvNeXt-TINY architecture itself (96/192/384/768 channels, depths 3/3/9/3: BN = 96 / 128 GEMM tiles,
# an 87.9 M-parameter arena beyond the 2^24 exact-fp32 index range, nine-block stage 2) -- backward vs autograd over the
# oracle, the full 384x384 x 64 step, and the two-stage top-down pipeline on ConvNeXt backbones.
# ------------------------------------------------------------------------------------------
def test_convnext_tiny_backward_matches_autograd():
    """model_type "tiny" forward + MSE + backward at 64x96, B = 2, 13 nodes: loss 1e-5 relative, every one of the 180+ parameter
    gradients within 2e-4 of its tensor's max magnitude (same bars as the toy architectures above)."""
    bb = _bb(model_type="tiny", output_stride=2)
    tm, *_ = _train_case(None, None, 2, 2, (64, 96), 2, bb=bb, layer_scale=0.3, n_parts=13, seed=41)
    assert tm.params.numel() > 2 ** 24 * 5  # 87.9 M parameters


def test_cfg4_full_size_training_properties():
    """The benched cfg4 workload at FULL size -- ConvNeXt-tiny centered-instance, 64 crops of 384x384, 13 nodes, output stride 2 -- through
    size-independent properties: every head output and gradient finite, the loss goes down over four Adam steps, the steps are
    bitwise repeatable from the same start, and crop 0 of the 64-crop inference forward equals the 1-crop forward of the same crop."""
    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.data.targets import generate_multiconfmaps
    from sleap_nn_amd.training.module import TrainingModule

    B, S = 64, 384
    bb = _bb(model_type="tiny", output_stride=2)
    heads = {"confmaps": {"part_names": [str(i) for i in range(13)], "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0, "anchor_part": None}}
    g = torch.Generator().manual_seed(4321)
    img = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, generator=g).to(DEV)
    rng = np.random.RandomState(5)
    pts = torch.from_numpy(np.clip(S / 2 + rng.normal(0, 40, size=(B, 1, 13, 2)), 8, S - 9).astype(np.float32)).to(DEV)
    tgt = {"CenteredInstanceConfmapsHead": generate_multiconfmaps(pts, (S, S), sigma=2.5, output_stride=2)}
    batch = {"image": img, **tgt}

    def run():
        m = Model("convnext", bb, heads, "centered_instance")
        m.init_xavier_(seed=1234, head_scale=0.05)
        tm = TrainingModule(m, DEV, lr=1e-4, loss_weights=[1.0])  # bench.py's optimizer (Adam at 1e-3 overshoots on 88 M xavier-initialised parameters)
        losses = [tm.training_step(batch).clone() for _ in range(4)]
        torch.cuda.synchronize()
        assert bool(torch.isfinite(tm.grads).all()) and bool(torch.isfinite(tm.params).all())
        return [float(l[0]) for l in losses], tm.params.clone(), tm, m

    l1, p1, tm, m = run()
    assert all(np.isfinite(l1)) and l1[3] < l1[0], l1
    tm.close()
    del tm, m
    torch.cuda.empty_cache()
    l2, p2, tm, m = run()
    assert l1 == l2 and torch.equal(p1, p2)  # fixed-order reductions everywhere: bitwise repeatable
    # inference program on the trained weights: batch invariance of crop 0
    tm.model.load_flat_params(tm.params)
    tm.close()
    m.bind_live_params(None)
    m.eval().to(DEV)
    m.set_option("conv_wino4", 3)  # (the F(4x4,3x3) / F(2x2,3x3) choice depends on the tile count, i.e. on the batch: one kernel choice for the bitwise comparison)
    m.set_option("conv_splitk", 0)  # (and so does the split-K form of either kernel)
    o64 = m(img)["CenteredInstanceConfmapsHead"][:1].clone()
    o1 = m(img[:1])["CenteredInstanceConfmapsHead"]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o64).all()) and torch.equal(o64, o1)


def test_two_stage_topdown_on_convnext_backbones_matches_oracle():
    """cfg4 at inference time: TopDownLayer with a ConvNeXt-tiny centroid model and a ConvNeXt-tiny centered-instance model --
    centroid confmaps -> local peaks (top-k) -> bit-exact uint8 crops -> centered-instance confmaps -> global peaks -> image
    coordinates -- against the oracle's centroid_postprocess / topdown_stage2 on the same frames and weights."""
    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.inference.backends import HipBackend
    from sleap_nn_amd.inference.layers import CenteredInstanceLayer, CentroidLayer, PostprocessConfig, TopDownLayer

    crop = 96
    bb = _bb(model_type="tiny", output_stride=2)
    hc = {"confmaps": {"anchor_part": None, "sigma": 2.5, "output_stride": 2}}
    hi = {"confmaps": {"part_names": [str(i) for i in range(5)], "anchor_part": None, "sigma": 2.5, "output_stride": 2}}
    sdc = O.init_state_convnext(bb, hc, "centroid", seed=51, head_scale=1.0, layer_scale=0.3, randomize_affine=True)
    sdi = O.init_state_convnext(bb, hi, "centered_instance", seed=52, head_scale=1.0, layer_scale=0.3, randomize_affine=True)
    # frames: blobs on a noisy background so that the (random-weight) centroid map has well separated maxima
    rng = np.random.RandomState(3)
    H, W = 192, 256
    yy, xx = np.mgrid[0:H, 0:W]
    frames = []
    for b in range(2):
        f = rng.uniform(0, 40, size=(H, W))
        for _ in range(4):
            cy, cx = rng.uniform(40, H - 40), rng.uniform(40, W - 40)
            f += 200 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 9.0 ** 2))
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    img = torch.from_numpy(np.stack(frames))[:, None]
    ref_c = O.model_forward(sdc, bb, hc, "centroid", img, backbone="convnext")["CentroidConfmapsHead"]
    thr_c = float(ref_c.mean() + 2.0 * ref_c.std())
    cp, cv = O.centroid_postprocess(ref_c, 2, thr_c, "integral", 5, max_instances=4)
    assert int((~torch.isnan(cp[..., 0])).sum()) >= 3, "test frames must yield a few centroids"
    fwd_i = lambda c: O.model_forward(sdi, bb, hi, "centered_instance", c, backbone="convnext")["CenteredInstanceConfmapsHead"]
    valid = ~torch.isnan(cp).any(-1)
    probe = fwd_i(O.crop_bboxes(img, O.make_centered_bboxes(cp[valid], crop, crop), valid.nonzero()[:, 0]))
    thr_i = float(probe.amax(dim=(2, 3)).min()) * 0.5  # every node of every crop is above threshold: no NaN pattern to flip
    # stage 2 without sub-pixel refinement: random-weight confidence maps take both signs, which makes the integral offsets (ratios of
    # 25-term sums that may cancel) arbitrarily ill-conditioned -- the refinement arithmetic has its own tests (peaks.npz); this one is
    # about the two-stage pipeline: crops, the ConvNeXt forward on them, the global arg-max and the crop offset
    kp, kv, crops, bboxes, pts = O.topdown_stage2(img, cp, (crop, crop), fwd_i, 2, thr_i, None, 5)

    mc = Model("convnext", bb, hc, "centroid")
    mc.load_state_dict(sdc, strict=True)
    mi = Model("convnext", bb, hi, "centered_instance")
    mi.load_state_dict(sdi, strict=True)
    cl = CentroidLayer(HipBackend(mc, DEV), 2, max_instances=4, max_stride=32, postprocess_config=PostprocessConfig(peak_threshold=thr_c, max_instances=4))
    il = CenteredInstanceLayer(HipBackend(mi, DEV), 2, max_stride=32, postprocess_config=PostprocessConfig(peak_threshold=thr_i, refinement="none"))
    raw = cl.backend(img)["CentroidConfmapsHead"].cpu()
    scale = max(1.0, float(ref_c.abs().max()))
    assert float((raw - ref_c).abs().max()) / scale <= ATOL
    td = TopDownLayer(cl, il, (crop, crop), return_crops=True)
    out = td.predict(img)
    assert np.allclose(out.pred_centroids.cpu().numpy(), cp.numpy(), atol=1e-3, equal_nan=True)
    assert np.allclose(out.pred_centroid_values.cpu().numpy(), cv.numpy(), atol=ATOL * scale, equal_nan=True)
    idx = valid.nonzero().numpy()
    got_crops = out.crops.cpu().numpy()[idx[:, 0], idx[:, 1]]
    assert np.array_equal(got_crops, crops.numpy())  # bit-exact uint8 crops
    k = out.pred_keypoints.cpu().numpy()
    assert np.array_equal(np.isnan(k), np.isnan(kp.numpy()))
    assert np.allclose(k, kp.numpy(), atol=1e-3, equal_nan=True)
    si = max(1.0, float(probe.abs().max()))
    assert np.allclose(out.pred_peak_values.cpu().numpy(), kv.numpy(), atol=ATOL * si, equal_nan=True)
