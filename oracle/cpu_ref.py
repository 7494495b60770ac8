"""CPU oracle: a torch-CPU / numpy restatement of the sleap-nn hot path.

TEST INFRASTRUCTURE ONLY -- imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product (``sleap_nn_amd``) never imports it.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
vectors produced by running the reference itself in the build container
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``) and against the known-answer
vectors of the reference's own tests (tests/inference/test_peak_finding.py,
tests/inference/test_paf_grouping.py, tests/inference/parity_golden/bottomup.pkl).

Each function cites the reference file:line (relative to the sleap-nn repo) it restates.
Arithmetic is plain fp32 ATen-CPU / numpy, written independently of the reference code.
Third-party pieces the reference itself delegates to are used as-is:
``scipy.optimize.linear_sum_assignment`` (reference: inference/ops/paf.py:589).
"""

from __future__ import annotations

import math
from collections import deque
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

# --------------------------------------------------------------------------------------
# Network plan (architectures/unet.py:49-253, encoder_decoder.py:38-730, model.py:157-261)
# --------------------------------------------------------------------------------------

HEAD_ORDER = {
    # model.py:92-148
    "single_instance": [("SingleInstanceConfmapsHead", "confmaps")],
    "centroid": [("CentroidConfmapsHead", "confmaps")],
    "centered_instance": [("CenteredInstanceConfmapsHead", "confmaps")],
    "bottomup": [("MultiInstanceConfmapsHead", "confmaps"), ("PartAffinityFieldsHead", "pafs")],
    "multi_class_bottomup": [("MultiInstanceConfmapsHead", "confmaps"), ("ClassMapsHead", "class_maps")],
    "multi_class_topdown": [("CenteredInstanceConfmapsHead", "confmaps"), ("ClassVectorsHead", "class_vectors")],
}


def head_channels(head_name: str, cfg: dict) -> int:
    """heads.py:96-98,168-170,281-283,342-344,398-400."""
    if head_name == "CentroidConfmapsHead":
        return 1
    if head_name == "PartAffinityFieldsHead":
        return 2 * len(cfg["edges"])
    if head_name == "ClassMapsHead":
        return len(cfg["classes"])
    return len(cfg["part_names"])


def unet_plan(bb: dict) -> dict:
    """Enumerate the conv layers of the reference UNet with checkpoint-compatible names.

    unet.py:230-253 (from_config), :96-226 (blocks), encoder_decoder.py:274-316 (encoder),
    :634-703 (decoder).  Only ``stacks == 1`` is meaningful in the reference.
    """
    filters = int(bb["filters"])
    rate = bb["filters_rate"]
    k = int(bb.get("kernel_size", 3))
    stem_stride = bb.get("stem_stride", None)
    stem_blocks = int(round(math.log2(stem_stride))) if stem_stride else 0
    down = int(round(math.log2(bb["max_stride"]))) - stem_blocks
    up = int(round(math.log2(bb["max_stride"] / bb["output_stride"]))) + stem_blocks
    cpb = int(bb.get("convs_per_block", 2))
    middle = bool(bb.get("middle_block", True))
    interp = bool(bb.get("up_interpolate", True))
    cin = int(bb["in_channels"])
    assert stem_blocks <= 1, "stem_stride > 2 does not run in the reference (unet.py:287-288 vs encoder_decoder.py:652-676)"
    assert int(bb.get("stacks", 1)) == 1

    stem = []  # StemBlock (encoder_decoder.py:144-225): 7x7 convs (stem_kernel_size = 7, never configured), pool before convs from block 1 on
    prev = cin
    for b in range(stem_blocks):
        f = int(filters * (rate**b))
        stem.append({"pool": b > 0, "convs": [(f"backbone.stem.stem_stack.{b}.blocks.stem{b}_conv{i}", prev if i == 0 else f, f) for i in range(cpb)]})
        prev = f
    enc = []  # list of blocks: dict(pool=bool, convs=[(name, cin, cout)])
    for b in range(down):
        f = int(filters * (rate ** (b + stem_blocks)))
        convs = []
        for i in range(cpb):
            convs.append((f"backbone.encoders.0.encoder_stack.{b}.blocks.stack0_enc{b}_conv{i}", prev if i == 0 else f, f))
        enc.append({"pool": b + stem_blocks > 0, "convs": convs})
        prev = f
    mid = []
    enc_num = down + 1  # encoder_stack has down blocks + last pool
    fmid = int(filters * (rate ** (down + stem_blocks)))
    mb = 0
    if middle:
        if cpb > 1:
            convs = []
            for i in range(cpb - 1):
                convs.append((f"backbone.middle_blocks.{mb}.blocks.stack0_enc{enc_num}_middle_expand_conv{i}", prev if i == 0 else fmid, fmid))
            mid.append(convs)
            enc_num += 1
            mb += 1
            prev = fmid
        mid.append([(f"backbone.middle_blocks.{mb}.blocks.stack0_enc{enc_num}_middle_contract_conv0", fmid, fmid)])
        prev = fmid
    x_in = fmid  # decoder input channels (unet.py:198-206, block_contraction False)
    dec = []
    cur_stride = 2 ** (down + stem_blocks + (1 if stem_blocks else 0))  # unet.py:178-191: the stem's own final pool counts too
    stride_to_filters = {cur_stride: x_in}
    pin = x_in
    for b in range(up):
        fout = int(filters * (rate ** max(0, down + stem_blocks - 1 - b)))
        nxt = cur_stride // 2
        pfx = f"backbone.decoders.0.decoder_stack.{b}.blocks.stack0_dec{b}_s{cur_stride}_to_s{nxt}"
        blk = {"interp": interp, "skip_c": fout, "convs": [], "stride": nxt}
        if not interp:
            blk["trans"] = (pfx + "_trans_conv", pin, fout)
            first_in = fout + fout
        else:
            first_in = pin + fout
        for i in range(cpb):
            blk["convs"].append((pfx + f"_refine_conv{i}", first_in if i == 0 else fout, fout))
        dec.append(blk)
        stride_to_filters[nxt] = fout
        pin = fout
        cur_stride = nxt
    return {"stem": stem, "enc": enc, "mid": mid, "dec": dec, "stride_to_filters": stride_to_filters, "k": k, "down": down}


def init_state(bb: dict, head_cfgs: dict, model_type: str, seed: int = 1234, head_scale: float = 0.05) -> Dict[str, torch.Tensor]:
    """Synthetic weights: xavier-uniform convs, zero bias (training/utils.py:72-78), heads x0.05.

    BASELINE.md section 3 synthetic-input definition.
    """
    plan = unet_plan(bb)
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    k = plan["k"]

    def xavier(shape, fan_in, fan_out):
        bound = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def add_conv(name, ci, co, kk):
        sd[name + ".weight"] = xavier((co, ci, kk, kk), ci * kk * kk, co * kk * kk)
        sd[name + ".bias"] = torch.zeros(co)

    for blk in plan["stem"]:
        for n, ci, co in blk["convs"]:
            add_conv(n, ci, co, 7)
    for blk in plan["enc"]:
        for n, ci, co in blk["convs"]:
            add_conv(n, ci, co, k)
    for convs in plan["mid"]:
        for n, ci, co in convs:
            add_conv(n, ci, co, k)
    for blk in plan["dec"]:
        if not blk["interp"]:
            n, ci, co = blk["trans"]
            sd[n + ".weight"] = xavier((ci, co, 3, 3), co * 9, ci * 9)
            sd[n + ".bias"] = torch.zeros(co)
        for n, ci, co in blk["convs"]:
            add_conv(n, ci, co, k)
    for i, (hname, key) in enumerate(HEAD_ORDER[model_type]):
        hc = head_cfgs[key]
        ci = plan["stride_to_filters"][hc["output_stride"]]
        co = head_channels(hname, hc)
        sd[f"head_layers.{i}.{hname}.0.weight"] = xavier((co, ci, 1, 1), ci, co) * head_scale
        sd[f"head_layers.{i}.{hname}.0.bias"] = torch.zeros(co)
    return sd


CONVNEXT_ARCHS = {
    # convnext.py:187-192
    "tiny": {"depths": [3, 3, 9, 3], "channels": [96, 192, 384, 768]},
    "small": {"depths": [3, 3, 27, 3], "channels": [96, 192, 384, 768]},
    "base": {"depths": [3, 3, 27, 3], "channels": [128, 256, 512, 1024]},
    "large": {"depths": [3, 3, 27, 3], "channels": [192, 384, 768, 1536]},
}
LN_EPS = 1e-6  # convnext.py:67 (LayerNorm2d) and torchvision CNBlock's nn.LayerNorm


def convnext_plan(bb: dict) -> dict:
    """Layer enumeration of ConvNextWrapper with checkpoint-compatible names.

    convnext.py:19-130 (encoder: ``features`` = [stem, stage0, down0, stage1, down1, stage2, down2,
    stage3]), :133-361 (wrapper: extra max pool, middle expand/contract, Decoder with
    ``encoder_channels``); encoder_decoder.py:634-703 (decoder filters).  CNBlock / LayerNorm2d /
    Conv2dNormActivation are torchvision's (not vendored by the reference, torchvision is absent from
    this image): restated from the public definition -- dwconv7x7(groups=C) -> LayerNorm(C) ->
    Linear(C,4C) -> GELU -> Linear(4C,C), times ``layer_scale``, plus the input.  PARITY OF THE
    CNBlock ARITHMETIC IS UNPINNED (self-consistent only; cross-checked against HuggingFace transformers'
    independent ConvNeXt stage in tests/test_oracle_golden.py, which is not a torchvision pin); the wrapper's
    middle/decoder half is pinned against the reference's own Decoder / SimpleConvBlock.
    """
    mt = bb.get("model_type", None)
    arch = CONVNEXT_ARCHS[mt] if mt in CONVNEXT_ARCHS else (bb.get("arch", None) or CONVNEXT_ARCHS["tiny"])
    depths, channels = [int(d) for d in arch["depths"]], [int(c) for c in arch["channels"]]
    cin = int(bb.get("in_channels", 1))
    k = int(bb.get("kernel_size", 3))
    sk = int(bb.get("stem_patch_kernel", 4))
    ss = int(bb.get("stem_patch_stride", 2))
    rate = bb.get("filters_rate", 2)
    cpb = int(bb.get("convs_per_block", 2))
    interp = bool(bb.get("up_interpolate", True))
    contraction = bool(bb.get("block_contraction", False))
    os_ = int(bb["output_stride"])
    assert not contraction, "oracle covers block_contraction=False"
    max_stride = ss * 8 * 2
    up = int(math.log2(max_stride / (ss * os_))) + int(math.log2(ss))
    down = len(channels) - 1
    pfx = "backbone.enc.features"
    enc = [{"kind": "stem", "name": f"{pfx}.0", "cin": cin, "cout": channels[0], "k": sk, "stride": ss}]
    fi = 1
    for si, (d, c) in enumerate(zip(depths, channels)):
        enc.append({"kind": "stage", "blocks": [f"{pfx}.{fi}.{j}" for j in range(d)], "c": c})
        fi += 1
        if si + 1 < len(channels):
            enc.append({"kind": "down", "name": f"{pfx}.{fi}", "cin": c, "cout": channels[si + 1]})
            fi += 1
    last = channels[-1]
    fmid = int(last * rate)
    mid = []
    mb = 0
    if cpb > 1:
        mid.append([(f"backbone.middle_blocks.{mb}.blocks.convnext_middle_expand_conv{i}", last if i == 0 else fmid, fmid) for i in range(cpb - 1)])
        mb += 1
    mid.append([(f"backbone.middle_blocks.{mb}.blocks.convnext_middle_contract_conv0", fmid, fmid)])
    dec = []
    cur = max_stride
    stride_to_filters = {cur: fmid}
    pin = fmid
    enc_ch = channels[::-1]
    for b in range(up):
        fout = int(channels[0] * (rate ** max(0, down + 1 - 1 - b)))
        nxt = cur // 2
        name = f"backbone.dec.decoder_stack.{b}.blocks.dec{b}_s{cur}_to_s{nxt}"
        concat = not (b >= down + 1)  # encoder_decoder.py:659 (stem_blocks = 1)
        skip_c = enc_ch[b] if b < len(enc_ch) else fout
        blk = {"interp": interp, "skip_c": skip_c if concat else 0, "convs": [], "stride": nxt, "concat": concat}
        nconv = 2 if concat else 1  # convnext.py:288-301 does not forward convs_per_block: Decoder default 2
        if not interp:
            blk["trans"] = (name + "_trans_conv", pin, fout)
            first_in = fout + (skip_c if concat else 0)
        else:
            first_in = pin + (skip_c if concat else 0)
        for i in range(nconv):
            blk["convs"].append((name + f"_refine_conv{i}", first_in if i == 0 else fout, fout))
        dec.append(blk)
        stride_to_filters[nxt] = fout
        pin = fout
        cur = nxt
    return {"enc": enc, "mid": mid, "dec": dec, "stride_to_filters": stride_to_filters, "k": k, "channels": channels, "depths": depths, "max_stride": max_stride}


def init_state_convnext(bb: dict, head_cfgs: dict, model_type: str, seed: int = 1234, head_scale: float = 0.05, layer_scale: float = 1e-6,
                        randomize_affine: bool = False) -> Dict[str, torch.Tensor]:
    """Synthetic ConvNeXt weights: xavier-uniform Conv2d/Linear weights and zero biases
    (training/utils.py:72-78), LayerNorm weight 1 / bias 0, ``layer_scale`` constant (torchvision
    default 1e-6 via convnext.py:46).  ``randomize_affine`` perturbs biases / LayerNorm affine / layer
    scale so that parity tests exercise every parameter."""
    plan = convnext_plan(bb)
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def xavier(shape, fan_in, fan_out):
        bound = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def vec(n, base, spread):
        return torch.full((n,), float(base)) + ((torch.rand(n, generator=g) * 2 - 1) * spread if randomize_affine else 0.0)

    for e in plan["enc"]:
        if e["kind"] == "stem":
            n, ci, co, kk = e["name"], e["cin"], e["cout"], e["k"]
            sd[n + ".0.weight"] = xavier((co, ci, kk, kk), ci * kk * kk, co * kk * kk)
            sd[n + ".0.bias"] = vec(co, 0.0, 0.1)
            sd[n + ".1.weight"] = vec(co, 1.0, 0.3)
            sd[n + ".1.bias"] = vec(co, 0.0, 0.1)
        elif e["kind"] == "stage":
            c = e["c"]
            for n in e["blocks"]:
                sd[n + ".layer_scale"] = (vec(c, layer_scale, 0.5 * layer_scale)).reshape(c, 1, 1)
                sd[n + ".block.0.weight"] = xavier((c, 1, 7, 7), 49, c * 49)
                sd[n + ".block.0.bias"] = vec(c, 0.0, 0.1)
                sd[n + ".block.2.weight"] = vec(c, 1.0, 0.3)
                sd[n + ".block.2.bias"] = vec(c, 0.0, 0.1)
                sd[n + ".block.3.weight"] = xavier((4 * c, c), c, 4 * c)
                sd[n + ".block.3.bias"] = vec(4 * c, 0.0, 0.1)
                sd[n + ".block.5.weight"] = xavier((c, 4 * c), 4 * c, c)
                sd[n + ".block.5.bias"] = vec(c, 0.0, 0.1)
        else:
            n, ci, co = e["name"], e["cin"], e["cout"]
            sd[n + ".0.weight"] = vec(ci, 1.0, 0.3)
            sd[n + ".0.bias"] = vec(ci, 0.0, 0.1)
            sd[n + ".1.weight"] = xavier((co, ci, 2, 2), ci * 4, co * 4)
            sd[n + ".1.bias"] = vec(co, 0.0, 0.1)
    k = plan["k"]
    for convs in plan["mid"]:
        for n, ci, co in convs:
            sd[n + ".weight"] = xavier((co, ci, k, k), ci * k * k, co * k * k)
            sd[n + ".bias"] = vec(co, 0.0, 0.05)
    for blk in plan["dec"]:
        if not blk["interp"]:
            n, ci, co = blk["trans"]
            sd[n + ".weight"] = xavier((ci, co, 3, 3), co * 9, ci * 9)
            sd[n + ".bias"] = vec(co, 0.0, 0.05)
        for n, ci, co in blk["convs"]:
            sd[n + ".weight"] = xavier((co, ci, k, k), ci * k * k, co * k * k)
            sd[n + ".bias"] = vec(co, 0.0, 0.05)
    for i, (hname, key) in enumerate(HEAD_ORDER[model_type]):
        hc = head_cfgs[key]
        ci = plan["stride_to_filters"][hc["output_stride"]]
        co = head_channels(hname, hc)
        sd[f"head_layers.{i}.{hname}.0.weight"] = xavier((co, ci, 1, 1), ci, co) * head_scale
        sd[f"head_layers.{i}.{hname}.0.bias"] = torch.zeros(co)
    return sd


def layer_norm_2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torchvision LayerNorm2d: permute to NHWC, F.layer_norm over C, permute back."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, LN_EPS).permute(0, 3, 1, 2)


def cn_block(sd: Dict[str, torch.Tensor], n: str, x: torch.Tensor) -> torch.Tensor:
    """torchvision CNBlock.forward (stochastic depth p = 0: convnext.py:45 default, identity)."""
    c = x.shape[1]
    y = F.conv2d(x, sd[n + ".block.0.weight"], sd[n + ".block.0.bias"], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (c,), sd[n + ".block.2.weight"], sd[n + ".block.2.bias"], LN_EPS)
    y = F.linear(y, sd[n + ".block.3.weight"], sd[n + ".block.3.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[n + ".block.5.weight"], sd[n + ".block.5.bias"])
    y = y.permute(0, 3, 1, 2)
    return sd[n + ".layer_scale"] * y + x


def decoder_forward(sd, plan: dict, x: torch.Tensor, feats: List[torch.Tensor], collect: Optional[dict] = None):
    """encoder_decoder.py:522-558,705-730 (shared by the UNet and ConvNeXt wrappers)."""
    pad = plan["k"] // 2
    outs, strides = [], []
    for i, blk in enumerate(plan["dec"]):
        if blk["interp"]:
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        else:
            n, _, _ = blk["trans"]
            x = F.relu(F.conv_transpose2d(x, sd[n + ".weight"], sd[n + ".bias"], stride=2, padding=1, output_padding=1))
            if collect is not None:
                collect[n] = x
        if i < len(feats) and blk.get("concat", True):
            skf = feats[i]
            if x.shape[-2:] != skf.shape[-2:]:
                x = F.interpolate(x, size=skf.shape[-2:], mode="bilinear", align_corners=False)
            x = torch.cat((skf, x), dim=1)
        for n, _, _ in blk["convs"]:
            x = F.relu(F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], padding=pad))
            if collect is not None:
                collect[n] = x
        outs.append(x)
        strides.append(blk["stride"])
    return outs, strides


def convnext_forward(sd: Dict[str, torch.Tensor], bb: dict, x: torch.Tensor, collect: Optional[dict] = None) -> dict:
    """ConvNextWrapper.forward (convnext.py:334-361) over ConvNeXtEncoder._forward_impl (:112-117)."""
    plan = convnext_plan(bb)
    pad = plan["k"] // 2
    enc_out = []
    for e in plan["enc"]:
        if e["kind"] == "stem":
            n = e["name"]
            x = F.conv2d(x, sd[n + ".0.weight"], sd[n + ".0.bias"], stride=e["stride"], padding=1)
            x = layer_norm_2d(x, sd[n + ".1.weight"], sd[n + ".1.bias"])
        elif e["kind"] == "stage":
            for n in e["blocks"]:
                x = cn_block(sd, n, x)
                if collect is not None:
                    collect[n] = x
        else:
            n = e["name"]
            x = layer_norm_2d(x, sd[n + ".0.weight"], sd[n + ".0.bias"])
            x = F.conv2d(x, sd[n + ".1.weight"], sd[n + ".1.bias"], stride=2)
        if collect is not None and e["kind"] != "stage":
            collect[e["name"]] = x
        enc_out.append(x)
    feats = enc_out[::2][::-1]
    x = same_pool2(enc_out[-1])
    for convs in plan["mid"]:
        for n, _, _ in convs:
            x = F.relu(F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], padding=pad))
            if collect is not None:
                collect[n] = x
    middle = x
    outs, strides = decoder_forward(sd, plan, x, feats, collect)
    return {"outputs": outs, "strides": strides, "middle_output": middle}


def normalize_input(x: torch.Tensor) -> torch.Tensor:
    """data/normalization.py:7-35 + the n_samples squeeze of lightning_modules.py:1840-1848."""
    if x.dim() == 5:
        x = x.squeeze(1)
    if not torch.is_floating_point(x):
        return x.float() / 255.0
    x = x.float()
    if x.max() > 1.0:
        x = x / 255.0
    return x


def same_pool2(x: torch.Tensor) -> torch.Tensor:
    """common.py:69-107: zero pad bottom/right when odd, then max_pool2d(2, 2)."""
    h, w = x.shape[-2:]
    ph, pw = h % 2, w % 2
    if ph or pw:
        x = F.pad(x, (0, pw, 0, ph))
    return F.max_pool2d(x, 2, 2)


def unet_forward(sd: Dict[str, torch.Tensor], bb: dict, x: torch.Tensor, collect: Optional[dict] = None) -> dict:
    """unet.py:260-299 / encoder_decoder.py:318-336,522-558,705-730 as one functional pass."""
    plan = unet_plan(bb)
    pad = plan["k"] // 2

    def conv(name, t, p=pad):
        y = F.relu(F.conv2d(t, sd[name + ".weight"], sd[name + ".bias"], padding=p))
        if collect is not None:
            collect[name] = y
        return y

    stem_out = None
    for blk in plan["stem"]:
        if blk["pool"]:
            x = same_pool2(x)
        for n, _, _ in blk["convs"]:
            x = conv(n, x, 3)
    if plan["stem"]:
        x = same_pool2(x)
        stem_out = x
    feats = []
    for blk in plan["enc"]:
        if blk["pool"]:
            x = same_pool2(x)
        for n, _, _ in blk["convs"]:
            x = conv(n, x)
        feats.append(x)
    x = same_pool2(x)
    for convs in plan["mid"]:
        for n, _, _ in convs:
            x = conv(n, x)
    middle = x
    feats = feats[::-1]
    if stem_out is not None:
        feats.append(stem_out)  # unet.py:287-288
    outs, strides = [], []
    for i, blk in enumerate(plan["dec"]):
        if blk["interp"]:
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        else:
            n, _, _ = blk["trans"]
            x = F.relu(F.conv_transpose2d(x, sd[n + ".weight"], sd[n + ".bias"], stride=2, padding=1, output_padding=1))
            if collect is not None:
                collect[n] = x
        if i < len(feats):
            sk = feats[i]
            if x.shape[-2:] != sk.shape[-2:]:
                x = F.interpolate(x, size=sk.shape[-2:], mode="bilinear", align_corners=False)
            x = torch.cat((sk, x), dim=1)  # skip first (encoder_decoder.py:545,556)
        for n, _, _ in blk["convs"]:
            x = conv(n, x)
        outs.append(x)
        strides.append(blk["stride"])
    return {"outputs": outs, "strides": strides, "middle_output": middle}


def model_forward(sd, bb: dict, head_cfgs: dict, model_type: str, image: torch.Tensor, collect: Optional[dict] = None, backbone: str = "unet") -> Dict[str, torch.Tensor]:
    """model.py:237-261 with the LightningModule forward's normalisation in front."""
    x = normalize_input(image)
    # float64 weights select a float64 evaluation (tolerance studies only: how far is fp32 autograd itself from the exact
    # gradient?); with the reference's fp32 weights this is a no-op
    x = x.to(next(iter(sd.values())).dtype)
    cin = int(bb["in_channels"])
    if x.shape[-3] != cin:
        if x.shape[-3] == 1:
            x = x.repeat(1, 3, 1, 1)
        elif x.shape[-3] == 3:
            # torchvision rgb_to_grayscale weights
            r, g, b = x.unbind(dim=-3)
            x = (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)
    bo = convnext_forward(sd, bb, x, collect) if backbone == "convnext" else unet_forward(sd, bb, x, collect)
    out = {}
    for i, (hname, key) in enumerate(HEAD_ORDER[model_type]):
        hc = head_cfgs[key]
        if hname == "ClassVectorsHead":
            # heads.py:506-539 on the decoder's input feature (model.py:197-199,253-255): global max pool ->
            # (Linear + ReLU) x num_fc_layers -> Linear -> softmax over classes
            v = F.adaptive_max_pool2d(bo["middle_output"], 1).flatten(1)
            for j in range(int(hc.get("num_fc_layers", 1))):
                v = F.relu(F.linear(v, sd[f"head_layers.{i}.pre_classification{j}_fc.weight"], sd[f"head_layers.{i}.pre_classification{j}_fc.bias"]))
            out[hname] = torch.softmax(F.linear(v, sd[f"head_layers.{i}.ClassVectorsHead.weight"], sd[f"head_layers.{i}.ClassVectorsHead.bias"]), dim=-1)
            continue
        feat = bo["outputs"][bo["strides"].index(hc["output_stride"])] if bo["outputs"] else bo["middle_output"]
        y = F.conv2d(feat, sd[f"head_layers.{i}.{hname}.0.weight"], sd[f"head_layers.{i}.{hname}.0.bias"])
        if hname == "ClassMapsHead":
            y = torch.sigmoid(y)
        out[hname] = y
    return out


def conv_flops(bb: dict, head_cfgs: dict, model_type: str, h: int, w: int) -> float:
    """Algorithmic forward FLOPs/frame: 2*Cin*Cout*k^2*Hout*Wout per conv."""
    plan = unet_plan(bb)
    k2 = plan["k"] ** 2
    total = 0.0
    ch, cw = h, w
    for blk in plan["stem"]:
        if blk["pool"]:
            ch, cw = (ch + 1) // 2, (cw + 1) // 2
        for _, ci, co in blk["convs"]:
            total += 2.0 * ci * co * 49 * ch * cw
    if plan["stem"]:
        ch, cw = (ch + 1) // 2, (cw + 1) // 2
    for blk in plan["enc"]:
        if blk["pool"]:
            ch, cw = (ch + 1) // 2, (cw + 1) // 2
        for _, ci, co in blk["convs"]:
            total += 2.0 * ci * co * k2 * ch * cw
    ch, cw = (ch + 1) // 2, (cw + 1) // 2
    for convs in plan["mid"]:
        for _, ci, co in convs:
            total += 2.0 * ci * co * k2 * ch * cw
    sizes = {}
    for blk in plan["dec"]:
        ch, cw = ch * 2, cw * 2
        if not blk["interp"]:
            _, ci, co = blk["trans"]
            total += 2.0 * ci * co * 9 * (ch // 2) * (cw // 2)
        for _, ci, co in blk["convs"]:
            total += 2.0 * ci * co * k2 * ch * cw
        sizes[blk["stride"]] = (ch, cw)
    for hname, key in HEAD_ORDER[model_type]:
        hc = head_cfgs[key]
        s = hc["output_stride"]
        hh, ww = sizes[s]
        total += 2.0 * plan["stride_to_filters"][s] * head_channels(hname, hc) * hh * ww
    return total


# --------------------------------------------------------------------------------------
# Peak finding (inference/ops/peaks.py, ops/crops.py, data/instance_cropping.py:129-171)
# --------------------------------------------------------------------------------------


def _neighbour_max(cms: torch.Tensor) -> torch.Tensor:
    """peaks.py:26-63: max over the 8 neighbours, -inf outside the image."""
    p = F.pad(cms, (1, 1, 1, 1), value=float("-inf"))
    h, w = cms.shape[-2:]
    best = None
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if dy == 1 and dx == 1:
                continue
            s = p[..., dy : dy + h, dx : dx + w]
            best = s if best is None else torch.maximum(best, s)
    return best


def local_peaks_rough(cms: torch.Tensor, threshold: float = 0.2):
    """peaks.py:184-218.  Order = lexicographic (sample, y, x, channel)."""
    mask = (cms > _neighbour_max(cms)) & (cms > threshold)
    b, y, x, c = torch.nonzero(mask.permute(0, 2, 3, 1), as_tuple=True)
    vals = cms[b, c, y, x]
    pts = torch.stack([x, y], dim=1).to(torch.float32)
    return pts, vals, b.to(torch.int32), c.to(torch.int32)


def _integral_offsets(cms_flat: torch.Tensor, pts: torch.Tensor, map_inds: torch.Tensor, patch: int) -> torch.Tensor:
    """Zero-padded patch around each integer peak, first moments / mass.

    instance_cropping.py:129-171 (bbox = centre -/+ (patch-1)/2), crops.py:85-124 (top-left
    ``trunc(tl + half) - half``; zero outside), peaks.py:66-86 (sum(g*crop)/sum(crop)).
    ``cms_flat`` is (M, H, W).
    """
    n = pts.shape[0]
    h, w = cms_flat.shape[-2:]
    half = patch // 2
    tl = pts - (patch - 1) / 2.0  # (n,2) x,y of top-left
    tl = (tl + half).to(torch.long) - half
    ar = torch.arange(patch)
    xs = tl[:, 0:1] + ar[None, :]  # (n,p)
    ys = tl[:, 1:2] + ar[None, :]
    okx = (xs >= 0) & (xs < w)
    oky = (ys >= 0) & (ys < h)
    xc = xs.clamp(0, w - 1)
    yc = ys.clamp(0, h - 1)
    crops = cms_flat[map_inds.long()[:, None, None], yc[:, :, None], xc[:, None, :]]
    crops = crops * (oky[:, :, None] & okx[:, None, :]).to(crops.dtype)
    crops = crops.reshape(n, 1, patch, patch)
    gv = torch.arange(patch, dtype=torch.float32) - (patch - 1) / 2
    z = torch.sum(crops, dim=[2, 3])
    dx = torch.sum(gv.view(1, 1, 1, -1) * crops, dim=[2, 3]) / z
    dy = torch.sum(gv.view(1, 1, -1, 1) * crops, dim=[2, 3]) / z
    return torch.cat([dx, dy], dim=1)


def find_local_peaks(cms: torch.Tensor, threshold: float = 0.2, refinement: Optional[str] = None, integral_patch_size: int = 5):
    """peaks.py:221-259."""
    pts, vals, sb, sc = local_peaks_rough(cms, threshold)
    if pts.shape[0] == 0 or refinement != "integral":
        return pts, vals, sb, sc
    bsz, ch = cms.shape[:2]
    flat = cms.reshape(bsz * ch, cms.shape[2], cms.shape[3])
    off = _integral_offsets(flat, pts, sb.long() * ch + sc.long(), integral_patch_size)
    return pts + off, vals, sb, sc


def global_peaks_rough(cms: torch.Tensor, threshold: float = 0.1):
    """peaks.py:89-130: x = first column holding the max, y = first row holding it."""
    colmax, _ = torch.max(cms, dim=2)  # (B,C,W): max over rows
    vmax, xi = torch.max(colmax, dim=2)
    rowmax, _ = torch.max(cms, dim=3)  # (B,C,H)
    _, yi = torch.max(rowmax, dim=2)
    pts = torch.stack([xi, yi], dim=-1).to(torch.float32)
    below = vmax < threshold
    pts = torch.where(below[..., None], torch.full_like(pts, float("nan")), pts)
    vals = torch.where(below, torch.zeros_like(vmax), vmax)
    return pts, vals


def find_global_peaks(cms: torch.Tensor, threshold: float = 0.2, refinement: Optional[str] = None, integral_patch_size: int = 5):
    """peaks.py:133-181."""
    pts, vals = global_peaks_rough(cms, threshold)
    if refinement != "integral" or torch.isnan(pts).all():
        return pts, vals
    b, c = cms.shape[:2]
    flat_pts = pts.reshape(b * c, 2)
    valid = torch.nonzero(~torch.isnan(flat_pts[:, 0]), as_tuple=True)[0]
    off = _integral_offsets(cms.reshape(b * c, cms.shape[2], cms.shape[3]), flat_pts[valid], valid, integral_patch_size)
    out = flat_pts.clone()
    out[valid] += off
    return out.reshape(b, c, 2), vals


# --------------------------------------------------------------------------------------
# PAF scoring (inference/ops/paf.py:84-497, inference/utils.py:29-130)
# --------------------------------------------------------------------------------------


def connection_candidates(chan: torch.Tensor, edges: Sequence[Tuple[int, int]], n_nodes: int):
    """paf.py:84-130: per edge, all (src peak, dst peak) pairs, src-major."""
    chan = chan.long()
    per_node = [torch.nonzero(chan == k, as_tuple=True)[0] for k in range(n_nodes)]
    e_inds, pairs = [], []
    for k, (s, d) in enumerate(edges):
        a, b = per_node[s], per_node[d]
        pr = torch.stack([a.repeat_interleave(b.numel()), b.repeat(a.numel())], dim=1)
        pairs.append(pr)
        e_inds.append(torch.full((pr.shape[0],), k, dtype=torch.int32))
    if not pairs:
        return torch.zeros(0, dtype=torch.int32), torch.zeros(0, 2, dtype=torch.long)
    return torch.cat(e_inds), torch.cat(pairs)


def line_points(src: torch.Tensor, dst: torch.Tensor, n_points: int, stride: int, hw: Tuple[int, int]):
    """paf.py:133-234 + utils.py:29-130 for x=[0,1]: y0 + ((y1-y0)/(eps+1)) * t.

    Returns int rows, cols of shape (n, n_points).
    """
    t = torch.linspace(0, 1, steps=n_points)
    den = torch.tensor(torch.finfo(torch.float32).eps) + 1  # fp32: 1 + 2^-23
    sl = (dst - src) / den  # (n,2)
    xy = src[:, :, None] + sl[:, :, None] * t[None, None, :]  # (n,2,P)
    ij = (xy / stride).round().int()
    rows = ij[:, 1].clamp(0, hw[0] - 1)
    cols = ij[:, 0].clamp(0, hw[1] - 1)
    return rows, cols


def score_lines_sample(pafs_hwc: torch.Tensor, peaks: torch.Tensor, e_inds: torch.Tensor, pairs: torch.Tensor, n_points: int, stride: int, max_edge_length: float, dist_penalty_weight: float):
    """paf.py:237-287 (gather) + :290-410 (score)."""
    if pairs.shape[0] == 0:
        return torch.zeros(0)
    src = peaks[pairs[:, 0]].float()
    dst = peaks[pairs[:, 1]].float()
    rows, cols = line_points(src, dst, n_points, stride, pafs_hwc.shape[:2])
    ch = e_inds.long()[:, None] * 2
    fx = pafs_hwc[rows.long(), cols.long(), ch]
    fy = pafs_hwc[rows.long(), cols.long(), ch + 1]
    vec = dst - src
    length = torch.norm(vec, dim=1, keepdim=True)
    unit = vec / length
    lines = torch.stack([fx, fy], dim=-1)  # (n,P,2)
    dots = torch.squeeze(lines @ unit.unsqueeze(2), dim=-1)
    pen = torch.clamp(max_edge_length / length - 1, max=0) * dist_penalty_weight
    return dots.mean(dim=1) + pen.squeeze(1)


def score_paf_lines_batch(pafs_bhwc: torch.Tensor, peaks: List[torch.Tensor], chans: List[torch.Tensor], edges, n_points: int, stride: int, max_edge_length_ratio: float, dist_penalty_weight: float, n_nodes: int):
    """paf.py:413-497.  Note max() includes the channel dim (paf.py:457-461)."""
    mel = max_edge_length_ratio * max(pafs_bhwc.shape[-1], pafs_bhwc.shape[-2], pafs_bhwc.shape[-3]) * stride
    oe, op, os_ = [], [], []
    for b in range(pafs_bhwc.shape[0]):
        e, p = connection_candidates(chans[b], edges, n_nodes)
        s = score_lines_sample(pafs_bhwc[b], peaks[b], e, p, n_points, stride, mel, dist_penalty_weight)
        oe.append(e)
        op.append(p)
        os_.append(s)
    return oe, op, os_


# --------------------------------------------------------------------------------------
# Matching + assembly (paf.py:500-1149)
# --------------------------------------------------------------------------------------


def toposort_edges(edges: Sequence[Tuple[int, int]]) -> Tuple[int, ...]:
    """paf.py:890-912 without networkx: root = first node of a topological order
    (networkx's generator yields zero-in-degree nodes in insertion order, so the first is
    the first-inserted node with in-degree 0), then BFS edge order from it with neighbours
    in insertion order."""
    order: List[int] = []
    adj: Dict[int, List[int]] = {}
    indeg: Dict[int, int] = {}
    seen_edges = set()
    for s, d in edges:
        for n in (s, d):
            if n not in adj:
                adj[n] = []
                indeg[n] = 0
                order.append(n)
        if (s, d) not in seen_edges:
            seen_edges.add((s, d))
            adj[s].append(d)
            indeg[d] += 1
    if not order:
        return tuple()
    roots = [n for n in order if indeg[n] == 0]
    if not roots:
        raise ValueError("skeleton graph has a cycle")
    root = roots[0]
    out = []
    visited = {root}
    q = deque([root])
    edge_list = [tuple(e) for e in edges]
    while q:
        u = q.popleft()
        for v in adj[u]:
            if v not in visited:
                visited.add(v)
                out.append(edge_list.index((u, v)))
                q.append(v)
    return tuple(out)


def match_candidates_sample(e_inds: torch.Tensor, pairs: torch.Tensor, scores: torch.Tensor, n_edges: int):
    """paf.py:500-619: per edge Hungarian on -score; NaN -> +inf."""
    e = e_inds.numpy()
    pr = pairs.numpy()
    sc = scores.numpy().astype(np.float32)
    me, ms, md, msc = [], [], [], []
    for k in range(n_edges):
        sel = np.nonzero(e == k)[0]
        pk = pr[sel]
        sk = sc[sel]
        su = np.unique(pk[:, 0]) if sel.size else np.zeros(0, dtype=np.int64)
        du = np.unique(pk[:, 1]) if sel.size else np.zeros(0, dtype=np.int64)
        cost = np.full((su.size, du.size), np.inf, dtype=np.float32)
        if sel.size:
            cost[np.searchsorted(su, pk[:, 0]), np.searchsorted(du, pk[:, 1])] = -sk
        cost[np.isnan(cost)] = np.inf
        r, c = linear_sum_assignment(cost)
        me.append(np.full(r.size, k, dtype=np.int32))
        ms.append(r.astype(np.int32))
        md.append(c.astype(np.int32))
        msc.append((-cost[r, c]).astype(np.float32))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt)
    return cat(me, np.int32), cat(ms, np.int32), cat(md, np.int32), cat(msc, np.float32)


def assemble_instances(peaks: np.ndarray, vals: np.ndarray, chans: np.ndarray, me, ms, md, msc, n_nodes: int, edges, sorted_edge_inds, min_instance_peaks=0, min_line_scores: float = 0.25):
    """paf.py:915-1038 (group_instances_sample) + :705-820 + :823-887."""
    keep = msc >= min_line_scores
    me, ms, md, msc = me[keep], ms[keep], md[keep], msc[keep]
    node_peaks = [peaks[chans == i] for i in range(n_nodes)]
    node_vals = [vals[chans == i] for i in range(n_nodes)]
    conns = []  # (edge_ind, [(src, dst, score)])
    for ei in sorted_edge_inds:
        sel = me == ei
        conns.append((ei, list(zip(ms[sel].tolist(), md[sel].tolist(), msc[sel]))))
    assign: Dict[Tuple[int, int], int] = {}
    for ei, lst in conns:
        sn, dn = edges[ei]
        for s, d, _ in lst:
            a, b = (sn, s), (dn, d)
            ia, ib = assign.get(a), assign.get(b)
            if ia is None and ib is None:
                nid = max(assign.values(), default=-1) + 1
                assign[a] = nid
                assign[b] = nid
            elif ia is not None and ib is None:
                assign[b] = ia
            elif ia is not None and ib is not None:
                assign[b] = ia
                na = {p[0] for p, v in assign.items() if v == ia}
                nb = {p[0] for p, v in assign.items() if v == ib}
                if not (na & nb):
                    for p in assign:
                        if assign[p] == ib:
                            assign[p] = ia
            # src unassigned & dst assigned: no-op
    if min_instance_peaks > 0:
        mip = min_instance_peaks
        if isinstance(mip, float):
            mip = int(mip * n_nodes)
        ids, counts = np.unique(list(assign.values()), return_counts=True)
        cnt = dict(zip(ids.tolist(), counts.tolist()))
        assign = {p: v for p, v in assign.items() if cnt[v] >= mip}
    ids, inv = np.unique(list(assign.values()), return_inverse=True)
    for p, j in zip(list(assign.keys()), inv):
        assign[p] = int(j)
    n_inst = len(ids)
    inst_scores = np.zeros(n_inst, dtype=np.float32)
    for ei, lst in conns:
        sn, _ = edges[ei]
        for s, _, score in lst:
            j = assign.get((sn, s))
            if j is not None:
                inst_scores[j] += np.float32(score)
    inst = np.full((n_inst, n_nodes, 2), np.nan, dtype=np.float32)
    inst_vals = np.full((n_inst, n_nodes), np.nan, dtype=np.float32)
    for (node, pi), j in assign.items():
        inst[j, node] = node_peaks[node][pi]
        inst_vals[j, node] = node_vals[node][pi]
    return inst, inst_vals, inst_scores


class PAFScorerRef:
    """paf.py:1152-1532 value bundle (defaults at :1207-1213)."""

    def __init__(self, part_names, edges, pafs_stride, max_edge_length_ratio=0.25, dist_penalty_weight=1.0, n_points=10, min_instance_peaks=0, min_line_scores=0.25):
        self.part_names = list(part_names)
        self.edges = [tuple(e) for e in edges]
        self.pafs_stride = pafs_stride
        self.max_edge_length_ratio = max_edge_length_ratio
        self.dist_penalty_weight = dist_penalty_weight
        self.n_points = n_points
        self.min_instance_peaks = min_instance_peaks
        self.min_line_scores = min_line_scores
        self.edge_inds = [(self.part_names.index(s), self.part_names.index(d)) for s, d in self.edges]
        self.n_nodes = len(self.part_names)
        self.n_edges = len(self.edges)
        self.sorted_edge_inds = toposort_edges(self.edge_inds)

    def predict(self, pafs_bhwc, peaks, vals, chans):
        e, p, s = score_paf_lines_batch(pafs_bhwc, peaks, chans, self.edge_inds, self.n_points, self.pafs_stride, self.max_edge_length_ratio, self.dist_penalty_weight, self.n_nodes)
        out = []
        for b in range(len(peaks)):
            m = match_candidates_sample(e[b], p[b], s[b], self.n_edges)
            out.append(assemble_instances(peaks[b].numpy(), vals[b].numpy(), chans[b].numpy(), *m, self.n_nodes, self.edge_inds, self.sorted_edge_inds, self.min_instance_peaks, self.min_line_scores))
        return out, (e, p, s)


def bottomup_postprocess(cms: torch.Tensor, pafs: torch.Tensor, scorer: PAFScorerRef, cms_stride: int, peak_threshold: float = 0.2, refinement: Optional[str] = "integral", patch: int = 5, max_instances: Optional[int] = None, input_scale: float = 1.0, eff_scale: Optional[torch.Tensor] = None, max_peaks_per_node: Optional[int] = None):
    """layers/bottomup.py:95-236 + streaming.py:147-255 -> NaN-padded (B,I,N,2),(B,I,N),(B,I)."""
    pts, vals, sb, sc = find_local_peaks(cms, peak_threshold, refinement, patch)
    pts = pts * cms_stride
    bsz, n_nodes = cms.shape[:2]
    pk, pv, pc = [], [], []
    for b in range(bsz):
        m = sb == b
        pk.append(pts[m])
        pv.append(vals[m].float())
        pc.append(sc[m])
    skip = False
    if max_peaks_per_node is not None:
        for c in pc:
            if c.numel() and int(torch.bincount(c.long(), minlength=n_nodes).max()) > max_peaks_per_node:
                skip = True
    if skip:
        mi = max_instances or 1
        return (np.full((bsz, mi, n_nodes, 2), np.nan, np.float32), np.full((bsz, mi, n_nodes), np.nan, np.float32), np.full((bsz, mi), np.nan, np.float32))
    res, _ = scorer.predict(pafs.permute(0, 2, 3, 1), pk, pv, pc)
    insts = [r[0] for r in res]
    if input_scale != 1.0:
        insts = [i / np.float32(input_scale) for i in insts]
    if eff_scale is not None and not bool(torch.all(eff_scale == 1.0)):
        insts = [i / np.float32(eff_scale[b]) for b, i in enumerate(insts)]
    mi = max_instances or max((i.shape[0] for i in insts), default=0)
    if mi == 0:
        mi = 1
    k = np.full((bsz, mi, n_nodes, 2), np.nan, np.float32)
    v = np.full((bsz, mi, n_nodes), np.nan, np.float32)
    s = np.full((bsz, mi), np.nan, np.float32)
    for b in range(bsz):
        ib, vb, sb_ = insts[b], res[b][1], res[b][2]
        if max_instances is not None and ib.shape[0] > mi:
            order = np.argsort(sb_)[::-1]
            ib, vb, sb_ = ib[order], vb[order], sb_[order]
        n = min(ib.shape[0], mi)
        k[b, :n], v[b, :n], s[b, :n] = ib[:n], vb[:n], sb_[:n]
    return k, v, s


def single_instance_postprocess(cms: torch.Tensor, output_stride: int, peak_threshold: float = 0.2, refinement: Optional[str] = "integral", patch: int = 5, input_scale: float = 1.0, eff_scale: Optional[torch.Tensor] = None):
    """layers/single_instance.py:72-108 + ops/coord.py:27-76."""
    pts, vals = find_global_peaks(cms, peak_threshold, refinement, patch)
    if output_stride != 1:
        pts = pts * output_stride
    if input_scale != 1.0:
        pts = pts / input_scale
    if eff_scale is not None and not bool(torch.all(eff_scale == 1.0)):
        pts = pts / eff_scale.view(-1, 1, 1)
    return pts.unsqueeze(1), vals.unsqueeze(1)


# --------------------------------------------------------------------------------------
# Synthetic rendered heads for measurement (data/confidence_maps.py:96-166,
# data/edge_maps.py:15-220 semantics)
# --------------------------------------------------------------------------------------


def render_instances(size: int, n_nodes: int, n_inst: int, seed: int) -> np.ndarray:
    rng = np.random.RandomState(seed)
    centres = rng.uniform(150, size - 150, size=(n_inst, 1, 2))
    pts = centres + rng.normal(0, 40, size=(n_inst, n_nodes, 2))
    return np.clip(pts, 8, size - 9).astype(np.float32)


def render_confmaps(pts: np.ndarray, size: int, stride: int, sigma: float) -> torch.Tensor:
    """Per-node max over instances of unit Gaussians sampled on the stride grid."""
    g = torch.arange(0, size, stride, dtype=torch.float32)
    p = torch.from_numpy(pts)  # (I,N,2)
    dx = (g[None, None, :] - p[:, :, 0:1]) ** 2  # (I,N,W)
    dy = (g[None, None, :] - p[:, :, 1:2]) ** 2
    cm = torch.exp(-(dy[:, :, :, None] + dx[:, :, None, :]) / (2 * sigma**2))
    return cm.max(dim=0)[0]


def render_pafs(pts: np.ndarray, edges, size: int, stride: int, sigma: float) -> torch.Tensor:
    """Summed edge fields: unit vector weighted by exp(-d^2/(2 sigma^2)), d = point-segment distance."""
    g = torch.arange(0, size, stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(g, g, indexing="ij")
    grid = torch.stack([xx, yy], dim=-1)  # (H,W,2)
    p = torch.from_numpy(pts)
    out = torch.zeros(len(edges) * 2, g.numel(), g.numel())
    for e, (s, d) in enumerate(edges):
        for i in range(p.shape[0]):
            a, b = p[i, s], p[i, d]
            ab = b - a
            l2 = float(ab @ ab) + 1e-12
            t = ((grid - a) @ ab / l2).clamp(0, 1)
            proj = a + t[..., None] * ab
            dist2 = ((grid - proj) ** 2).sum(-1)
            wgt = torch.exp(-dist2 / (2 * sigma**2))
            u = ab / math.sqrt(l2)
            out[2 * e] += wgt * u[0]
            out[2 * e + 1] += wgt * u[1]
    return out


# --------------------------------------------------------------------------------------
# Top-down glue (layers/centroid.py:195-261, layers/topdown.py:183-300,
# layers/centered_instance.py:199-230, ops/crops.py:31-124, data/instance_cropping.py:129-171)
# --------------------------------------------------------------------------------------


def make_centered_bboxes(centroids: torch.Tensor, box_height: int, box_width: int) -> torch.Tensor:
    """instance_cropping.py:129-171: corners TL, TR, BR, BL of a box centred on each point."""
    hw, hh = box_width / 2, box_height / 2
    x, y = centroids[..., 0], centroids[..., 1]
    tl = torch.stack([x - hw, y - hh], -1)
    tr = torch.stack([x + hw, y - hh], -1)
    br = torch.stack([x + hw, y + hh], -1)
    bl = torch.stack([x - hw, y + hh], -1)
    corners = torch.stack([tl, tr, br, bl], dim=-2)
    return corners + torch.tensor([[0.5, 0.5], [-0.5, 0.5], [-0.5, -0.5], [0.5, -0.5]])


def crop_bboxes(images: torch.Tensor, bboxes: torch.Tensor, sample_inds: torch.Tensor) -> torch.Tensor:
    """crops.py:31-124: zero-padded integer crops; origin = trunc(tl + half) - half."""
    n = bboxes.shape[0]
    if n == 0:
        return torch.empty(0, images.shape[1], 0, 0, dtype=images.dtype)
    h = int(abs(bboxes[0, 3, 1] - bboxes[0, 0, 1]).item()) + 1
    w = int(abs(bboxes[0, 1, 0] - bboxes[0, 0, 0]).item()) + 1
    half = torch.tensor([w // 2, h // 2], dtype=bboxes.dtype)
    org = (bboxes[:, 0, :] + half).to(torch.long) - half.long()
    H, W = images.shape[-2:]
    out = torch.zeros((n, images.shape[1], h, w), dtype=images.dtype)
    for k in range(n):
        ox, oy = int(org[k, 0]), int(org[k, 1])
        x0, x1 = max(ox, 0), min(ox + w, W)
        y0, y1 = max(oy, 0), min(oy + h, H)
        if x1 > x0 and y1 > y0:
            out[k, :, y0 - oy : y1 - oy, x0 - ox : x1 - ox] = images[int(sample_inds[k]), :, y0:y1, x0:x1]
    return out


def centroid_postprocess(cms: torch.Tensor, output_stride: int, peak_threshold: float = 0.2, refinement: Optional[str] = "integral", patch: int = 5, max_instances: Optional[int] = None, input_scale: float = 1.0, eff_scale: Optional[torch.Tensor] = None):
    """layers/centroid.py:195-261 -> (B, I, 2) NaN-padded centroids, (B, I) values (top-k by value)."""
    pts, vals, sb, _ = find_local_peaks(cms, peak_threshold, refinement, patch)
    if output_stride != 1:
        pts = pts * output_stride
    if input_scale != 1.0:
        pts = pts / input_scale
    B = cms.shape[0]
    mi = max_instances or (int(torch.bincount(sb.long()).max()) if sb.numel() else 0)
    if mi == 0:
        mi = 1
    cp = torch.full((B, mi, 2), float("nan"))
    cv = torch.full((B, mi), float("nan"))
    for b in range(B):
        m = sb == b
        p, v = pts[m], vals[m]
        if p.numel() == 0:
            continue
        if p.shape[0] > mi:
            v, idx = torch.topk(v, mi)
            p = p[idx]
        cp[b, : p.shape[0]] = p
        cv[b, : p.shape[0]] = v
    if eff_scale is not None and not bool(torch.all(eff_scale == 1.0)):
        cp = cp / eff_scale.view(-1, 1, 1)
    return cp, cv


def topdown_stage2(image: torch.Tensor, centroids: torch.Tensor, crop_hw, forward_fn, output_stride: int, peak_threshold: float = 0.2, refinement: Optional[str] = "integral", patch: int = 5):
    """layers/topdown.py:183-300 with eff_scale == 1: crop valid centroids, run the centered-instance
    model (``forward_fn(crops uint8) -> confmaps``), global peaks, add the crop offset."""
    B, I, _ = centroids.shape
    valid = ~torch.isnan(centroids).any(dim=-1)
    idx = torch.nonzero(valid, as_tuple=False)
    vc = centroids[idx[:, 0], idx[:, 1]]
    bboxes = make_centered_bboxes(vc, crop_hw[0], crop_hw[1])
    crops = crop_bboxes(image, bboxes, idx[:, 0])
    cms = forward_fn(crops)
    pts, vals = find_global_peaks(cms, peak_threshold, refinement, patch)
    if output_stride != 1:
        pts = pts * output_stride
    n_nodes = pts.shape[1]
    kp = torch.full((B, I, n_nodes, 2), float("nan"))
    kv = torch.full((B, I, n_nodes), float("nan"))
    kp[idx[:, 0], idx[:, 1]] = pts + bboxes[:, 0, :].view(-1, 1, 2)
    kv[idx[:, 0], idx[:, 1]] = vals
    return kp, kv, crops, bboxes, pts


# --------------------------------------------------------------------------------------
# Multi-class bottom-up (ops/identity.py:13-146, layers/bottomup_multiclass.py:76-150)
# --------------------------------------------------------------------------------------


def group_class_peaks(probs: torch.Tensor, sb: torch.Tensor, sc: torch.Tensor, n_samples: int, n_channels: int):
    """identity.py:13-76."""
    pi_l, ci_l = [], []
    for s in range(n_samples):
        for c in range(n_channels):
            members = torch.nonzero((sb == s) & (sc == c), as_tuple=True)[0]
            if members.numel() == 0:
                continue
            r, k = linear_sum_assignment(-probs[members].numpy())
            pi_l.append(members[torch.from_numpy(r)])
            ci_l.append(torch.from_numpy(k).long())
    if not pi_l:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)
    pi, ci = torch.cat(pi_l), torch.cat(ci_l)
    keep = probs[pi, ci] == probs[pi].max(dim=1).values
    return pi[keep], ci[keep]


def classify_peaks_from_maps(class_maps: torch.Tensor, pts: torch.Tensor, vals: torch.Tensor, sb: torch.Tensor, sc: torch.Tensor, n_channels: int):
    """identity.py:79-146."""
    B, K, H, W = class_maps.shape
    rc = torch.round(pts).to(torch.int32)
    col = rc[:, 0].clamp(0, W - 1).long()
    row = rc[:, 1].clamp(0, H - 1).long()
    probs = class_maps[sb.long(), :, row, col]
    pi, ci = group_class_peaks(probs, sb, sc, B, n_channels)
    points = torch.full((B, K, n_channels, 2), float("nan"))
    pvals = torch.full((B, K, n_channels), float("nan"))
    cprobs = torch.full((B, K, n_channels), float("nan"))
    points[sb[pi].long(), ci, sc[pi].long()] = pts[pi]
    pvals[sb[pi].long(), ci, sc[pi].long()] = vals[pi]
    cprobs[sb[pi].long(), ci, sc[pi].long()] = probs[pi, ci]
    return points, pvals, cprobs


def multiclass_postprocess(cms: torch.Tensor, class_maps: torch.Tensor, cms_stride: int, cm_stride: int, peak_threshold: float = 0.2, refinement: Optional[str] = "integral", patch: int = 5, input_scale: float = 1.0, eff_scale: Optional[torch.Tensor] = None):
    """layers/bottomup_multiclass.py:76-127 (no max_instances cap)."""
    pts, vals, sb, sc = find_local_peaks(cms, peak_threshold, refinement, patch)
    pts = pts * cms_stride
    inst, pvals, cprobs = classify_peaks_from_maps(class_maps, pts / cm_stride, vals, sb, sc, cms.shape[1])
    inst = inst * cm_stride
    if input_scale != 1.0:
        inst = inst / input_scale
    if eff_scale is not None and not bool(torch.all(eff_scale == 1.0)):
        inst = inst / eff_scale.view(-1, 1, 1, 1)
    return inst, pvals, torch.nanmean(pvals, dim=-1), torch.nanmean(cprobs, dim=-1)


# --------------------------------------------------------------------------------------
# Training step (training/lightning_modules.py:490-545,1850-1922, training/losses.py:8-63,
# torch.optim.Adam at lightning_modules.py:752-763) -- torch autograd on the CPU.
# --------------------------------------------------------------------------------------


def ohkm_loss(y_gt: torch.Tensor, y_pr: torch.Tensor, hard_to_easy_ratio=2.0, min_hard_keypoints=2, max_hard_keypoints=None, loss_scale=5.0) -> torch.Tensor:
    """losses.py:8-63."""
    loss = (y_pr - y_gt) ** 2
    shp = loss.shape
    per_ch = torch.sum(loss, dim=(0, 2, 3))
    best = torch.min(per_ch)
    n_hard = int(torch.sum(((per_ch / best) >= hard_to_easy_ratio).to(torch.int32)))
    mx = per_ch.shape[0] if (max_hard_keypoints is None or max_hard_keypoints < 0) else min(max_hard_keypoints, per_ch.shape[0])
    k = min(max(n_hard, min_hard_keypoints), mx)
    k_vals, _ = torch.topk(per_ch, k=k, largest=True, sorted=False)
    return torch.sum(k_vals * loss_scale) / (shp[0] * shp[2] * shp[3] * k)


def negative_weighted_mse(y_pr: torch.Tensor, y_gt: torch.Tensor, is_negative: Optional[torch.Tensor], negative_loss_weight: float = 1.0, stage: str = "train") -> torch.Tensor:
    """lightning_modules.py:490-545: plain nn.MSELoss without ``is_negative``; else the mean over samples of the per-sample
    MSE, negatives weighted by ``negative_loss_weight`` in the train stage only."""
    if is_negative is None:
        return F.mse_loss(y_pr, y_gt)
    per_sample = (y_pr - y_gt).pow(2).mean(dim=list(range(1, y_pr.ndim)))
    if stage != "train" or negative_loss_weight == 1.0:
        return per_sample.mean()
    w = torch.where(is_negative.bool(), torch.tensor(float(negative_loss_weight), dtype=per_sample.dtype), torch.tensor(1.0, dtype=per_sample.dtype))
    return (per_sample * w).mean()


def training_step(sd: Dict[str, torch.Tensor], bb: dict, head_cfgs: dict, model_type: str, image: torch.Tensor, targets: Dict[str, torch.Tensor],
                  loss_weights: Sequence[float], ohkm: Optional[dict] = None, backbone: str = "unet", is_negative: Optional[torch.Tensor] = None,
                  negative_loss_weight: float = 1.0, stage: str = "train"):
    """Forward + weighted per-head MSE (+OHKM) + autograd backward.  Returns (losses [total, heads...], grads dict)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = model_forward(params, bb, head_cfgs, model_type, image, backbone=backbone)
    heads = [h for h, _ in HEAD_ORDER[model_type]]
    hl = []
    for h in heads:
        if h == "ClassVectorsHead":  # lightning_modules.py:2655-2662: CrossEntropyLoss on the (already soft-maxed) head output
            hl.append(F.cross_entropy(out[h], targets[h]))
            continue
        l = negative_weighted_mse(out[h], targets[h], is_negative, negative_loss_weight, stage)
        if ohkm:
            l = l + ohkm_loss(targets[h], out[h], **ohkm)
        hl.append(l)
    total = sum(w * l for w, l in zip(loss_weights, hl))
    total.backward()
    return [float(total.detach())] + [float(l.detach()) for l in hl], {k: p.grad for k, p in params.items()}


def adam_reference(sd: Dict[str, torch.Tensor], grads_per_step: List[Dict[str, torch.Tensor]], lr=1e-4, amsgrad=False, optimizer: str = "Adam"):
    """torch.optim.Adam / AdamW themselves (lightning_modules.py:752-763), stepped with the given gradients."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = (torch.optim.AdamW if optimizer == "AdamW" else torch.optim.Adam)(list(params.values()), lr=lr, amsgrad=amsgrad)
    for g in grads_per_step:
        for k, p in params.items():
            p.grad = g[k].clone()
        opt.step()
    return {k: p.detach() for k, p in params.items()}


# --------------------------------------------------------------------------------------
# Training targets (data/confidence_maps.py:36-166, data/edge_maps.py:15-323, data/utils.py:55-125)
# --------------------------------------------------------------------------------------


def make_multiconfmaps(points: torch.Tensor, img_hw, sigma: float, stride: int) -> torch.Tensor:
    """(B, I, N, 2) -> (B, N, h, w): max over instances of the unit Gaussian, NaN points ignored."""
    xv = torch.arange(0, img_hw[1], step=stride, dtype=torch.float32)
    yv = torch.arange(0, img_hw[0], step=stride, dtype=torch.float32)
    s = sigma * stride
    B, I, N, _ = points.shape
    out = torch.zeros((B, N, yv.numel(), xv.numel()))
    for i in range(I):
        x = points[:, i, :, 0].reshape(B, N, 1, 1)
        y = points[:, i, :, 1].reshape(B, N, 1, 1)
        cm = torch.exp(-((xv.view(1, 1, 1, -1) - x) ** 2 + (yv.view(1, 1, -1, 1) - y) ** 2) / (2 * s**2))
        out = torch.maximum(out, torch.nan_to_num(cm))
    return out


def make_pafs_sample(inst: torch.Tensor, edges, img_hw, sigma: float, stride: int) -> torch.Tensor:
    """One sample (I, N, 2) -> (2E, h, w), restating generate_pafs incl. its quirks (squared distance fed
    to the Gaussian, in-image instance filter, NaN -> 0, sum over instances)."""
    xv = torch.arange(0, img_hw[1], step=stride, dtype=torch.float32)
    yv = torch.arange(0, img_hw[0], step=stride, dtype=torch.float32)
    lim = torch.stack([xv[-1], yv[-1]]).view(1, 1, 2)
    keep = ((inst > 0) & (inst < lim)).all(dim=-1).any(dim=1)
    inst = inst[keep]
    yy, xx = torch.meshgrid(yv, xv, indexing="ij")
    grid = torch.stack((xx, yy), dim=-1)
    out = torch.zeros((len(edges) * 2, yv.numel(), xv.numel()))
    for i in range(inst.shape[0]):
        for e, (s, d) in enumerate(edges):
            a, b = inst[i, s], inst[i, d]
            v = b - a
            l2 = torch.maximum((v * v).sum(), torch.tensor(1.0))
            rel = grid - a
            t = ((rel * v).sum(-1) / l2).clamp(0, 1)
            d2 = ((t.unsqueeze(-1) * v - rel) ** 2).sum(-1)
            g = torch.exp(-(d2**2) / (2 * sigma**2))
            u = v / torch.norm(v)
            px, py = g * u[0], g * u[1]
            out[2 * e] += torch.nan_to_num(px, nan=0.0)
            out[2 * e + 1] += torch.nan_to_num(py, nan=0.0)
    return out


def class_inds_from_vectors(probs: torch.Tensor):
    """ops/identity.py:149-173: Hungarian matching of samples to classes on -prob; unmatched -> -1 / NaN."""
    r, c = linear_sum_assignment(-probs.numpy())
    inds = torch.full((probs.shape[0],), -1, dtype=torch.int64)
    pr = torch.full((probs.shape[0],), float("nan"))
    for a, b in zip(r, c):
        inds[a] = int(b)
        pr[a] = probs[a, b]
    return inds, pr


# ---------------------------------------------------------------------------------------------
# Preprocessing resizes (data/resizing.py:70-84 resize_image, :136-175 apply_sizematcher).  The reference calls
# torchvision.transforms.v2.functional.resize (bilinear, antialias=True), i.e. torch's interpolate(..., antialias=True);
# torchvision is not in the reference tree.  For uint8 frames (what the layers keep, layers/base.py:212-253) the arithmetic
# below restates ATen's separable int16 fixed-point kernel (aten/src/ATen/native/cpu/UpSampleKernel.cpp:
# _compute_indices_int16_weights_aa + the uint8 single-dim loop) and is pinned bit-exactly against torch's own CPU operator
# in tests/test_oracle_golden.py; float frames go through torch's operator directly.
# ---------------------------------------------------------------------------------------------
def _aa_axis_table(in_size: int, out_size: int):
    import math

    scale = in_size / out_size
    support = scale if scale >= 1.0 else 1.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    max_taps = int(math.ceil(support)) * 2 + 1
    start = np.zeros(out_size, np.int64)
    count = np.zeros(out_size, np.int64)
    w = np.zeros((out_size, max_taps), np.float64)
    for i in range(out_size):
        center = scale * (i + 0.5)
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        x = np.abs((np.arange(n) + lo - center + 0.5) * invscale)
        v = np.where(x < 1.0, 1.0 - x, 0.0)
        tot = 0.0
        for t in v:  # sequential sum, as the C loop does
            tot += t
        w[i, :n] = v / tot if tot != 0.0 else v
        start[i], count[i] = lo, n
    w_max = float(w.max())
    precision = 0
    while precision < 22:
        if int(0.5 + w_max * (1 << (precision + 1))) >= (1 << 15):
            break
        precision += 1
    wi = np.where(w < 0, (-0.5 + w * (1 << precision)).astype(np.int64), (0.5 + w * (1 << precision)).astype(np.int64))
    return start, count, wi, precision


def resize_bilinear_aa(image: torch.Tensor, size) -> torch.Tensor:
    """interpolate(image, size, mode="bilinear", align_corners=False, antialias=True) as torchvision's resize issues it."""
    oh, ow = int(size[0]), int(size[1])
    if image.dtype != torch.uint8:
        x = image if image.dim() == 4 else image[None]
        y = F.interpolate(x.float(), size=(oh, ow), mode="bilinear", align_corners=False, antialias=True)
        return y if image.dim() == 4 else y[0]
    x = image.numpy().astype(np.int64)
    H, W = x.shape[-2:]
    if ow != W:  # horizontal pass first, uint8 intermediate
        start, count, wi, prec = _aa_axis_table(W, ow)
        out = np.empty(x.shape[:-1] + (ow,), np.int64)
        for i in range(ow):
            acc = (1 << (prec - 1)) + (x[..., start[i] : start[i] + count[i]] * wi[i, : count[i]]).sum(-1)
            out[..., i] = np.clip(acc >> prec, 0, 255)
        x = out
    if oh != H:
        start, count, wi, prec = _aa_axis_table(H, oh)
        out = np.empty(x.shape[:-2] + (oh, x.shape[-1]), np.int64)
        for i in range(oh):
            acc = (1 << (prec - 1)) + (x[..., start[i] : start[i] + count[i], :] * wi[i, : count[i], None]).sum(-2)
            out[..., i, :] = np.clip(acc >> prec, 0, 255)
        x = out
    return torch.from_numpy(x.astype(np.uint8))


def resize_image(image: torch.Tensor, scale: float) -> torch.Tensor:
    """data/resizing.py:70-84."""
    h, w = image.shape[-2:]
    return resize_bilinear_aa(image, [int(h * scale), int(w * scale)])


def apply_sizematcher(image: torch.Tensor, max_height: Optional[int] = None, max_width: Optional[int] = None):
    """data/resizing.py:136-175: fit into (max_height, max_width) keeping the aspect ratio, zero pad bottom/right."""
    h, w = image.shape[-2:]
    max_height = h if max_height is None else max_height
    max_width = w if max_width is None else max_width
    if h == max_height and w == max_width:
        return image, 1.0
    hratio, wratio = max_height / h, max_width / w
    if hratio > wratio:
        eff, th, tw = wratio, int(round(h * wratio)), int(round(w * wratio))
    else:
        eff, tw, th = hratio, int(round(w * hratio)), int(round(h * hratio))
    image = resize_bilinear_aa(image, (th, tw))
    image = F.pad(image, (0, max_width - tw, 0, max_height - th), mode="constant")
    return image, eff


def full_preprocess(x: torch.Tensor, max_height=None, max_width=None, scale: float = 1.0, max_stride: int = 1):
    """layers/base.py:270-374 steps 2-4 on a (B, C, H, W) tensor -> (processed, eff_scale (B,), original (H, W))."""
    B, _c, H, W = x.shape
    if max_height is not None or max_width is not None:
        frames, effs = [], []
        for b in range(B):
            r, e = apply_sizematcher(x[b], max_height, max_width)
            frames.append(r)
            effs.append(float(e))
        x = torch.stack(frames, 0)
        eff = torch.tensor(effs, dtype=torch.float32)
    else:
        eff = torch.ones(B, dtype=torch.float32)
    if scale != 1.0:
        x = resize_image(x, scale)
    if max_stride != 1:
        h, w = x.shape[-2:]
        x = F.pad(x, (0, (max_stride - w % max_stride) % max_stride, 0, (max_stride - h % max_stride) % max_stride), mode="constant")
    return x, eff, (H, W)
