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
def _train_case(channels, depths, stem_stride, os_, hw, B, model_type="centered_instance", in_ch=1, seed=13):
    from sleap_nn_amd.architectures.model import Model
    from sleap_nn_amd.training.module import TrainingModule

    bb = _bb(arch={"depths": depths, "channels": channels}, stem_patch_stride=stem_stride, output_stride=os_, in_channels=in_ch)
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "sigma": 2.5, "output_stride": os_, "loss_weight": 1.0, "anchor_part": None}}
    sd = O.init_state_convnext(bb, heads, model_type, seed=seed, head_scale=1.0, layer_scale=0.6, randomize_affine=True)
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
    for step in range(3):
        _, grads = O.training_step({k: p.detach() for k, p in opt_params.items()}, bb, heads, "centered_instance", img, targets, [1.0], backbone="convnext")
        for k, p in opt_params.items():
            p.grad = grads[k]
        opt.step()
        loss = tm.training_step(batch)
        first = float(loss[0]) if first is None else first
    torch.cuda.synchronize()
    got = tm.state_dict()
    # an Adam step moves every parameter by at most ~lr whatever the gradient's size, so fp32 noise on tiny
    # gradients shows up at a fixed fraction of lr: allow 2 % of the total possible movement (3 steps x lr)
    for k, p in opt_params.items():
        assert float((got[k].cpu() - p.detach()).abs().max()) <= 0.02 * 3 * 1e-3, k
    assert float(loss[0]) < first
