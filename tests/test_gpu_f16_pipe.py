"""The round-6 kernels of the plain-fp16 precision (the reference's autocast mode, torch_backend.py:113-143; tolerance tests/inference/test_cuda.py:52-56 = 5e-3):
conv3x3_f16_rows_kernel (row tiles, loader waves, folded bilinear x2, fused pool / 1x1 heads), block2_c32_f16_kernel (two convs of an encoder block in one launch) and
stem_f16_kernel (both convs of the first block on the fp16 matrix pipe) -- each forced on / off against the other kernels of the same handle and against the fp32 oracle."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FP16_ATOL = 5e-3  # the reference's own fp16 bar (tests/inference/test_cuda.py:54-55)


def _net(filters, rate, max_stride, out_stride, heads_kind, nodes=5, in_ch=1):
    bb = {"in_channels": in_ch, "kernel_size": 3, "filters": filters, "filters_rate": rate, "max_stride": max_stride, "stem_stride": None, "middle_block": True, "up_interpolate": True,
          "stacks": 1, "convs_per_block": 2, "output_stride": out_stride}
    names = [f"n{i}" for i in range(nodes)]
    if heads_kind == "bottomup":
        heads = {"confmaps": {"part_names": names, "output_stride": out_stride}, "pafs": {"edges": [[names[i], names[i + 1]] for i in range(nodes - 1)], "output_stride": 2 * out_stride}}
        return bb, heads, "bottomup"
    if heads_kind == "multiclass":
        heads = {"confmaps": {"part_names": names, "output_stride": out_stride}, "class_maps": {"classes": ["a", "b", "c"], "output_stride": 2 * out_stride}}
        return bb, heads, "multi_class_bottomup"
    return bb, {"confmaps": {"part_names": names, "output_stride": out_stride}}, "single_instance"


def _run(sd, bb, heads, mt, img, opts, reuse=1):
    from sleap_nn_amd.architectures.model import Model

    m = Model("unet", bb, heads, mt)
    m.load_state_dict(sd)
    m.set_option("workspace_reuse", reuse)
    for k, v in opts.items():
        m.set_option(k, v)
    m.to(DEV).set_precision("fp16")
    out = {k: v.cpu().clone() for k, v in m(img.to(DEV)).items()}
    again = {k: v.cpu() for k, v in m(img.to(DEV)).items()}
    for k in out:
        assert torch.equal(out[k], again[k]), (k, "not repeatable")
    return out, m.last_kernels(), m


@pytest.mark.parametrize("filters,rate,max_stride,out_stride,kind,hw,batch", [
    (16, 2, 32, 4, "multiclass", (192, 192), 2),    # cfg5's network: 64- and 128-channel head producers, three folded bilinears, 96 / 48 / 24 / 12 / 6-pixel maps
    (16, 2, 8, 2, "bottomup", (104, 88), 3),        # maps of 104 x 88, 52 x 44, 26 x 22, 13 x 11: partial tiles, widths that are no multiple of 16 (per-lane fragment addresses)
    (24, 1.5, 8, 2, "single", (72, 120), 1),        # 24 / 36 / 54 / 81 channels: N tiles that are not full, channel counts padded to 32
    (16, 2, 16, 4, "bottomup", (256, 640), 1),      # wide maps: column strips of the row tiles
])
def test_row_tile_kernel_on_every_layer_it_takes_equals_the_persistent_kernel_and_the_oracle(filters, rate, max_stride, out_stride, kind, hw, batch):
    """conv3x3_f16_rows_kernel forced on every 3x3 conv whose shape it takes (conv_f16_rows = 2) vs the round-2 kernel (0) and the automatic routing (1): the same fp16
    operands and the same K order -> head outputs within a few fp32 roundings of each other (folded bilinear: the SAME packed-fp16 arithmetic as the standalone launch), all of
    them within the fp16 bar of the fp32 oracle; inference plan and keep-everything plan give the same bits."""
    from sleap_nn_amd import _lib as L

    bb, heads, mt = _net(filters, rate, max_stride, out_stride, kind)
    sd = O.init_state(bb, heads, mt, seed=filters + hw[0], head_scale=1.0)
    img = torch.randint(0, 256, (batch, 1, hw[0], hw[1]), dtype=torch.uint8, generator=torch.Generator().manual_seed(hw[1]))
    ref = O.model_forward(sd, bb, heads, mt, img)
    outs = {}
    for mode in (0, 1, 2):
        outs[mode], kv, m = _run(sd, bb, heads, mt, img, {"conv_f16_rows": mode, "block_fuse": 0})
        n_rows = sum(1 for c in kv if c == L.KV_F16_ROWS)
        assert (n_rows == 0) if mode == 0 else (n_rows >= 1 if mode == 2 else True), (mode, kv)
        for k, v in ref.items():
            assert (outs[mode][k] - v).abs().max().item() <= FP16_ATOL, (mode, k)
    for k, v in ref.items():
        scale = max(v.abs().max().item(), 1.0)
        for mode in (1, 2):
            assert (outs[mode][k] - outs[0][k]).abs().max().item() <= 2e-3 * scale, (mode, k)  # (an fp32 rounding of a sum can flip an fp16 rounding of an activation: a few fp16 ulps at the heads)
    keep, _kv, _m = _run(sd, bb, heads, mt, img, {"conv_f16_rows": 2, "block_fuse": 0}, reuse=0)  # nothing folded, nothing skipped: every tensor exists
    for k in ref:
        assert (keep[k] - outs[2][k]).abs().max().item() <= 2e-3 * max(ref[k].abs().max().item(), 1.0), k


def test_folded_bilinear_is_the_standalone_launch_bit_for_bit_in_both_arithmetics():
    """The bilinear x2 folded into conv3x3_f16_rows_kernel's loader waves vs upsample2x_fmt_kernel + the same conv on the up-sampled tensor: the SAME bits, in the packed-fp16
    arithmetic (upsample_f16math = 1, the default) and in the fp32 arithmetic (0); the two arithmetics differ by fp16 roundings of the up-sampled activations only."""
    bb, heads, mt = _net(16, 2, 16, 4, "bottomup")
    sd = O.init_state(bb, heads, mt, seed=77, head_scale=1.0)
    img = torch.randint(0, 256, (2, 1, 160, 96), dtype=torch.uint8, generator=torch.Generator().manual_seed(5))
    res = {}
    for math in (1, 0):
        folded, kv, _m = _run(sd, bb, heads, mt, img, {"conv_f16_rows": 2, "upsample_f16math": math, "upsample_fold": 1, "block_fuse": 0})
        apart, kv2, _m = _run(sd, bb, heads, mt, img, {"conv_f16_rows": 2, "upsample_f16math": math, "upsample_fold": 0, "block_fuse": 0})
        for k in folded:
            assert torch.equal(folded[k], apart[k]), (math, k)
        res[math] = folded
    ref = O.model_forward(sd, bb, heads, mt, img)
    for k, v in ref.items():
        assert (res[1][k] - v).abs().max().item() <= FP16_ATOL and (res[0][k] - v).abs().max().item() <= FP16_ATOL
        assert (res[1][k] - res[0][k]).abs().max().item() <= 2e-3 * max(v.abs().max().item(), 1.0)


@pytest.mark.parametrize("hw,batch,in_ch", [((96, 128), 2, 1), ((112, 80), 1, 1), ((64, 96), 2, 3), ((48, 272), 1, 1)])
def test_fused_encoder_block_and_fp16_stem_against_the_unfused_kernels_and_the_oracle(hw, batch, in_ch):
    """block2_c32_f16_kernel (conv 16 -> 32 + conv 32 -> 32 + pool in one launch, the intermediate in LDS) and stem_f16_kernel (first conv as an im2col product on the matrix
    cores, uint8 -> fp16 through a table) switched on / off: the library reports the fused launch, the outputs stay within the fp16 bar of the fp32 oracle and within a few
    fp16 roundings of the unfused kernels (the fused forms sum the taps in another order); image-cut tiles, odd sizes (the pool's zero padding), RGB input, float frames."""
    from sleap_nn_amd import _lib as L

    bb, heads, mt = _net(16, 2, 16, 4, "bottomup", in_ch=in_ch)
    sd = O.init_state(bb, heads, mt, seed=hw[0] + in_ch, head_scale=1.0)
    img = torch.randint(0, 256, (batch, in_ch, hw[0], hw[1]), dtype=torch.uint8, generator=torch.Generator().manual_seed(hw[1]))
    ref = O.model_forward(sd, bb, heads, mt, img)
    outs = {}
    for blk, stem in ((1, 1), (0, 1), (1, 0), (0, 0)):
        outs[(blk, stem)], kv, m = _run(sd, bb, heads, mt, img, {"block_fuse": blk, "stem_f16mfma": stem})
        assert (L.KV_F16_BLOCK in kv) == bool(blk), (blk, kv)
        for k, v in ref.items():
            assert (outs[(blk, stem)][k] - v).abs().max().item() <= FP16_ATOL, (blk, stem, k)
    for k, v in ref.items():
        scale = max(v.abs().max().item(), 1.0)
        for key in ((1, 1), (0, 1), (1, 0)):
            assert (outs[key][k] - outs[(0, 0)][k]).abs().max().item() <= 3e-3 * scale, (key, k)
    # normalised float frames take the arithmetic path of the stem (no table): the same bits as the uint8 frames (x / 255 is the same IEEE division on either side)
    of, _kv, _m = _run(sd, bb, heads, mt, img.float() / 255.0, {"block_fuse": 1, "stem_f16mfma": 1})
    for k in ref:
        assert torch.equal(of[k], outs[(1, 1)][k]), k
