"""ConvNeXt encoder + UNet-style decoder -> op program for libposehip.

Reproduces the structure and the checkpoint parameter names of the reference
``ConvNextWrapper`` (``sleap_nn/architectures/convnext.py:19-130`` encoder, ``:133-361`` wrapper,
decoder filters from ``encoder_decoder.py:634-703``) without torch modules.  The encoder blocks
are torchvision's (``CNBlock``, ``LayerNorm2d``, ``Conv2dNormActivation``); their state_dict
layout is kept (``features.{i}.{j}.block.{0,2,3,5}``, ``layer_scale``) so reference checkpoints load.

In the NHWC layout of the kernels the block is five launches: depthwise 7x7 -> LayerNorm over
channels -> row GEMM C->4C with the erf-GELU in its epilogue -> row GEMM 4C->C whose epilogue
applies ``layer_scale`` and adds the block input.  Stochastic depth is the identity (the wrapper
never sets a probability, convnext.py:45,212-217).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.unet import OpSpec
from sleap_nn_amd.utils import cfg_get

ARCH_TYPES = {
    # convnext.py:187-192
    "tiny": {"depths": [3, 3, 9, 3], "channels": [96, 192, 384, 768]},
    "small": {"depths": [3, 3, 27, 3], "channels": [96, 192, 384, 768]},
    "base": {"depths": [3, 3, 27, 3], "channels": [128, 256, 512, 1024]},
    "large": {"depths": [3, 3, 27, 3], "channels": [192, 384, 768, 1536]},
}


@dataclass
class ConvNextWrapper:
    """Config-derived description; ``from_config`` mirrors convnext.py:308-332."""

    model_type: str = "tiny"
    output_stride: int = 2
    depths: List[int] = field(default_factory=lambda: [3, 3, 9, 3])
    channels: List[int] = field(default_factory=lambda: [96, 192, 384, 768])
    in_channels: int = 1
    kernel_size: int = 3
    stem_patch_kernel: int = 4
    stem_patch_stride: int = 2
    filters_rate: float = 2
    convs_per_block: int = 2
    up_interpolate: bool = True
    block_contraction: bool = False
    ops: List[OpSpec] = field(default_factory=list)
    n_slots: int = 0
    decoder_stride_to_filters: Dict[int, int] = field(default_factory=dict)
    decoder_slot_of_stride: Dict[int, int] = field(default_factory=dict)
    middle_slot: int = -1
    param_shapes: Dict[str, Tuple[int, ...]] = field(default_factory=dict)
    labels: Dict[str, int] = field(default_factory=dict)

    @classmethod
    def from_config(cls, config) -> "ConvNextWrapper":
        mt = cfg_get(config, "model_type", None)
        arch = cfg_get(config, "arch", None)
        if mt in ARCH_TYPES:
            a = ARCH_TYPES[mt]
        elif arch is not None:
            a = {"depths": list(cfg_get(arch, "depths")), "channels": list(cfg_get(arch, "channels"))}
        else:
            a = ARCH_TYPES["tiny"]
        net = cls(
            model_type=mt,
            output_stride=int(cfg_get(config, "output_stride")),
            depths=[int(d) for d in a["depths"]],
            channels=[int(c) for c in a["channels"]],
            in_channels=int(cfg_get(config, "in_channels", 1)),
            kernel_size=int(cfg_get(config, "kernel_size", 3)),
            stem_patch_kernel=int(cfg_get(config, "stem_patch_kernel", 4)),
            stem_patch_stride=int(cfg_get(config, "stem_patch_stride", 2)),
            filters_rate=cfg_get(config, "filters_rate", 2),
            convs_per_block=int(cfg_get(config, "convs_per_block", 2)),
            up_interpolate=bool(cfg_get(config, "up_interpolate", True)),
            block_contraction=bool(cfg_get(config, "block_contraction", False)),
        )
        net._build()
        return net

    @property
    def max_stride(self) -> int:
        return self.stem_patch_stride * 8 * 2  # convnext.py:200-202

    @property
    def max_channels(self) -> int:
        return int(self.channels[-1] * self.filters_rate)

    # -- program construction ---------------------------------------------------------
    def _new_slot(self) -> int:
        self.n_slots += 1
        return self.n_slots - 1

    def _emit(self, op: OpSpec) -> int:
        self.ops.append(op)
        return op.dst

    def _conv3(self, name: str, src0: int, cin0: int, cout: int, src1: int = -1, cin1: int = 0) -> int:
        dst = self._new_slot()
        self.param_shapes[name + ".weight"] = (cout, cin0 + cin1, 3, 3)
        self.param_shapes[name + ".bias"] = (cout,)
        self.ops.append(OpSpec(L.OP_CONV, src0, src1, dst, cin0, cin1, cout, 3, L.FLAG_RELU, name + ".weight", name + ".bias", label=name))
        self.labels[name] = dst
        return dst

    def _layernorm(self, name: str, src: int, c: int) -> int:
        self.param_shapes[name + ".weight"] = (c,)
        self.param_shapes[name + ".bias"] = (c,)
        return self._emit(OpSpec(L.OP_LAYERNORM, src, -1, self._new_slot(), c, 0, c, 1, 0, name + ".weight", name + ".bias", label=name))

    def _cn_block(self, name: str, x: int, c: int) -> int:
        p = self.param_shapes
        p[name + ".layer_scale"] = (c, 1, 1)
        p[name + ".block.0.weight"] = (c, 1, 7, 7)
        p[name + ".block.0.bias"] = (c,)
        y = self._emit(OpSpec(L.OP_DWCONV, x, -1, self._new_slot(), c, 0, c, 7, 0, name + ".block.0.weight", name + ".block.0.bias", label=name + ".block.0"))
        y = self._layernorm(name + ".block.2", y, c)
        p[name + ".block.3.weight"] = (4 * c, c)
        p[name + ".block.3.bias"] = (4 * c,)
        # unfused form (what training runs: autograd needs the GELU input and the un-scaled block output);
        # Model._fuse_cnblocks folds GELU and layer-scale + residual into the GEMM epilogues for inference
        y = self._emit(OpSpec(L.OP_LINEAR, y, -1, self._new_slot(), c, 0, 4 * c, 1, 0, name + ".block.3.weight", name + ".block.3.bias", label=name + ".block.3"))
        y = self._emit(OpSpec(L.OP_GELU, y, -1, self._new_slot(), 4 * c, 0, 4 * c, 1, 0, label=name + ".block.4"))
        p[name + ".block.5.weight"] = (c, 4 * c)
        p[name + ".block.5.bias"] = (c,)
        y = self._emit(OpSpec(L.OP_LINEAR, y, -1, self._new_slot(), 4 * c, 0, c, 1, 0, name + ".block.5.weight", name + ".block.5.bias", label=name + ".block.5"))
        y = self._emit(OpSpec(L.OP_SCALE_ADD, y, x, self._new_slot(), c, 0, c, 1, 0, name + ".layer_scale", label=name))
        self.labels[name] = y
        return y

    def _build(self) -> None:
        if self.kernel_size != 3:
            raise ValueError("only kernel_size=3 is supported by the MFMA convolution kernels")
        if self.block_contraction:
            raise ValueError("block_contraction=True is not supported by the MI355X hot path yet")
        if self.stem_patch_stride not in (1, 2, 4) or not (2 <= self.stem_patch_kernel <= 8) or self.stem_patch_stride > self.stem_patch_kernel:
            raise ValueError("stem_patch_stride must be 1, 2 or 4 and stem_patch_kernel in 2..8")
        ch = self.channels
        pfx = "backbone.enc.features"
        # stem: conv k/s, padding 1 + LayerNorm2d (features.0)
        self.param_shapes[f"{pfx}.0.0.weight"] = (ch[0], self.in_channels, self.stem_patch_kernel, self.stem_patch_kernel)
        self.param_shapes[f"{pfx}.0.0.bias"] = (ch[0],)
        x = self._emit(OpSpec(L.OP_PATCH_STEM, -1, -1, self._new_slot(), self.in_channels, 0, ch[0], self.stem_patch_kernel, 0, f"{pfx}.0.0.weight",
                              f"{pfx}.0.0.bias", label=f"{pfx}.0.0", cmid=self.stem_patch_stride))
        x = self._layernorm(f"{pfx}.0.1", x, ch[0])
        self.labels[f"{pfx}.0"] = x
        skips: List[Tuple[int, int]] = [(x, ch[0])]  # enc_output[::2]: stem and downsample outputs
        fi = 1
        for si, (depth, c) in enumerate(zip(self.depths, ch)):
            for j in range(depth):
                x = self._cn_block(f"{pfx}.{fi}.{j}", x, c)
            fi += 1
            if si + 1 < len(ch):
                x = self._layernorm(f"{pfx}.{fi}.0", x, c)
                name = f"{pfx}.{fi}.1"
                self.param_shapes[name + ".weight"] = (ch[si + 1], c, 2, 2)
                self.param_shapes[name + ".bias"] = (ch[si + 1],)
                x = self._emit(OpSpec(L.OP_PATCH_CONV, x, -1, self._new_slot(), c, 0, ch[si + 1], 2, 0, name + ".weight", name + ".bias", label=name))
                self.labels[f"{pfx}.{fi}"] = x
                skips.append((x, ch[si + 1]))
                fi += 1
        cur, cur_c = x, ch[-1]
        # additional pool + middle blocks (convnext.py:219-270, 352-358)
        cur = self._emit(OpSpec(L.OP_POOL, cur, -1, self._new_slot(), cur_c, label="additional_pool"))
        fmid = int(ch[-1] * self.filters_rate)
        mb = 0
        if self.convs_per_block > 1:
            for i in range(self.convs_per_block - 1):
                cur = self._conv3(f"backbone.middle_blocks.{mb}.blocks.convnext_middle_expand_conv{i}", cur, cur_c, fmid)
                cur_c = fmid
            mb += 1
        cur = self._conv3(f"backbone.middle_blocks.{mb}.blocks.convnext_middle_contract_conv0", cur, fmid, fmid)
        cur_c = fmid
        self.middle_slot = cur
        # decoder (encoder_decoder.py:634-703 with stem_blocks=1, encoder_channels=channels[::-1]; the
        # wrapper does not forward convs_per_block, so refine blocks always have the Decoder default of 2)
        stride = self.max_stride
        ss, os_ = self.stem_patch_stride, self.output_stride
        up_blocks = int(math.log2(stride / (ss * os_))) + int(math.log2(ss))
        down_blocks = len(ch) - 1
        self.decoder_stride_to_filters = {stride: fmid}
        skips = skips[::-1]
        for b in range(up_blocks):
            fout = int(ch[0] * (self.filters_rate ** max(0, down_blocks + 1 - 1 - b)))
            nxt = stride // 2
            name = f"backbone.dec.decoder_stack.{b}.blocks.dec{b}_s{stride}_to_s{nxt}"
            concat = b < down_blocks + 1
            dst = self._new_slot()
            if self.up_interpolate:
                self.ops.append(OpSpec(L.OP_UPSAMPLE, cur, -1, dst, cur_c, label=name + "_interp_bilinear"))
                up_c = cur_c
            else:
                tn = name + "_trans_conv"
                self.param_shapes[tn + ".weight"] = (cur_c, fout, 3, 3)
                self.param_shapes[tn + ".bias"] = (fout,)
                self.ops.append(OpSpec(L.OP_CONVT, cur, -1, dst, cur_c, 0, fout, 3, L.FLAG_RELU, tn + ".weight", tn + ".bias", label=tn))
                self.labels[tn] = dst
                up_c = fout
            cur, cur_c = dst, up_c
            for i in range(2 if concat else 1):
                cn = name + f"_refine_conv{i}"
                if i == 0 and concat and b < len(skips):
                    sk, sk_c = skips[b]
                    cur = self._conv3(cn, sk, sk_c, fout, src1=cur, cin1=cur_c)  # concat (skip, x)
                else:
                    cur = self._conv3(cn, cur, cur_c, fout)
                cur_c = fout
            self.decoder_stride_to_filters[nxt] = fout
            self.decoder_slot_of_stride[nxt] = cur
            stride = nxt
