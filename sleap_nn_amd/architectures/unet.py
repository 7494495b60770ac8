"""UNet layer plan -> op program for libposehip.

Reproduces the *structure* and the checkpoint parameter names of the reference UNet
(``sleap_nn/architectures/unet.py:49-253`` and ``encoder_decoder.py:274-316,634-703``)
without any torch modules: the network is a flat list of ops over activation slots that the
C side (csrc/model.hip) executes with hand-written gfx950 kernels.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from sleap_nn_amd import _lib as L
from sleap_nn_amd.utils import cfg_get


@dataclass
class OpSpec:
    kind: int
    src0: int = -1
    src1: int = -1
    dst: int = -1
    cin0: int = 0
    cin1: int = 0
    cout: int = 0
    ksize: int = 3
    flags: int = 0
    weight: Optional[str] = None  # state_dict key
    bias: Optional[str] = None
    out_index: int = -1
    label: str = ""
    dst2: int = -1
    weight2: Optional[str] = None
    bias2: Optional[str] = None
    cmid: int = 0


@dataclass
class UNet:
    """Config-derived description; ``from_config`` mirrors unet.py:230-253."""

    in_channels: int = 1
    kernel_size: int = 3
    filters: int = 32
    filters_rate: float = 1.5
    down_blocks: int = 4
    up_blocks: int = 3
    stem_blocks: int = 0
    convs_per_block: int = 2
    middle_block: bool = True
    up_interpolate: bool = True
    stacks: int = 1
    output_stride: int = 2
    ops: List[OpSpec] = field(default_factory=list)
    n_slots: int = 0
    decoder_stride_to_filters: Dict[int, int] = field(default_factory=dict)
    decoder_slot_of_stride: Dict[int, int] = field(default_factory=dict)
    middle_slot: int = -1
    param_shapes: Dict[str, Tuple[int, ...]] = field(default_factory=dict)
    labels: Dict[str, int] = field(default_factory=dict)  # conv name -> output slot

    @classmethod
    def from_config(cls, config) -> "UNet":
        stem_stride = cfg_get(config, "stem_stride", None)
        stem_blocks = int(round(math.log2(stem_stride))) if stem_stride else 0
        max_stride = cfg_get(config, "max_stride")
        output_stride = cfg_get(config, "output_stride")
        net = cls(
            in_channels=int(cfg_get(config, "in_channels", 1)),
            kernel_size=int(cfg_get(config, "kernel_size", 3)),
            filters=int(cfg_get(config, "filters", 32)),
            filters_rate=cfg_get(config, "filters_rate", 1.5),
            down_blocks=int(round(math.log2(max_stride))) - stem_blocks,
            up_blocks=int(round(math.log2(max_stride / output_stride))) + stem_blocks,
            stem_blocks=stem_blocks,
            convs_per_block=int(cfg_get(config, "convs_per_block", 2)),
            middle_block=bool(cfg_get(config, "middle_block", True)),
            up_interpolate=bool(cfg_get(config, "up_interpolate", True)),
            stacks=int(cfg_get(config, "stacks", 1)),
            output_stride=int(output_stride),
        )
        net._build()
        return net

    # -- program construction ---------------------------------------------------------
    def _new_slot(self) -> int:
        self.n_slots += 1
        return self.n_slots - 1

    def _conv(self, name: str, src0: int, cin0: int, cout: int, src1: int = -1, cin1: int = 0, first: bool = False, k: Optional[int] = None) -> int:
        dst = self._new_slot()
        k = self.kernel_size if k is None else k
        self.param_shapes[name + ".weight"] = (cout, cin0 + cin1, k, k)
        self.param_shapes[name + ".bias"] = (cout,)
        self.ops.append(
            OpSpec(L.OP_INPUT_CONV if first else L.OP_CONV, src0, src1, dst, cin0, cin1, cout, k, L.FLAG_RELU, name + ".weight", name + ".bias", label=name)
        )
        self.labels[name] = dst
        return dst

    def _pool(self, cur: int, cur_c: int, label: str) -> int:
        dst = self._new_slot()
        self.ops.append(OpSpec(L.OP_POOL, cur, -1, dst, cur_c, label=label))
        return dst

    def _build(self) -> None:
        if self.stem_blocks > 1:
            # unet.py:287-288 appends ONE stem feature, the decoder (encoder_decoder.py:652-676) then builds its last blocks for
            # a concat that never arrives: the reference's forward fails with a channel mismatch for stem_stride >= 4
            raise ValueError("stem_stride > 2 does not run in the reference either (decoder / feature-list mismatch, unet.py:287-288)")
        if self.stacks != 1:
            raise ValueError("only stacks=1 is supported (the reference's multi-stack path is non-functional, unet.py:120,277)")
        if self.kernel_size % 2 != 1 or not (1 <= self.kernel_size <= 9):
            raise ValueError("kernel_size must be odd and <= 9")
        f, r, sb = self.filters, self.filters_rate, self.stem_blocks
        cur, cur_c = -1, self.in_channels
        stem_out: Optional[Tuple[int, int]] = None
        for b in range(sb):  # StemBlock (encoder_decoder.py:144-225): 7x7 convs (stem_kernel_size is never configured, unet.py:51), pool, ...
            bf = int(f * (r**b))
            if b > 0:
                cur = self._pool(cur, cur_c, f"stem{b}_pool")
            for i in range(self.convs_per_block):
                cur = self._conv(f"backbone.stem.stem_stack.{b}.blocks.stem{b}_conv{i}", cur, cur_c, bf, first=(b == 0 and i == 0), k=7)
                cur_c = bf
        if sb > 0:
            cur = self._pool(cur, cur_c, f"stem{sb}_last_pool")
            stem_out = (cur, cur_c)
        skips: List[Tuple[int, int]] = []
        for b in range(self.down_blocks):
            bf = int(f * (r ** (b + sb)))
            if b + sb > 0:
                cur = self._pool(cur, cur_c, f"stack0_enc{b}_pool")
            for i in range(self.convs_per_block):
                name = f"backbone.encoders.0.encoder_stack.{b}.blocks.stack0_enc{b}_conv{i}"
                cur = self._conv(name, cur, cur_c, bf, first=(b + sb == 0 and i == 0))
                cur_c = bf
            skips.append((cur, cur_c))
        cur = self._pool(cur, cur_c, f"stack0_enc{self.down_blocks}_last_pool")
        enc_num = self.down_blocks + 1
        fmid = int(f * (r ** (self.down_blocks + sb)))
        mb = 0
        if self.middle_block:
            if self.convs_per_block > 1:
                for i in range(self.convs_per_block - 1):
                    name = f"backbone.middle_blocks.{mb}.blocks.stack0_enc{enc_num}_middle_expand_conv{i}"
                    cur = self._conv(name, cur, cur_c, fmid)
                    cur_c = fmid
                enc_num += 1
                mb += 1
            name = f"backbone.middle_blocks.{mb}.blocks.stack0_enc{enc_num}_middle_contract_conv0"
            cur = self._conv(name, cur, fmid, fmid)
            cur_c = fmid
        elif cur_c != fmid:
            # The reference declares the decoder input (and max_channels) as filters * rate**down_blocks whether or not the
            # middle block exists (unet.py:196-205,256-258); without it the tensor that arrives has the last encoder block's
            # channel count, and the reference's first decoder conv fails on the mismatch at forward time.  Same contract here,
            # reported when the model is built.
            raise ValueError(
                f"middle_block=False leaves {cur_c} channels at the decoder input but the decoder is declared with {fmid} "
                "(unet.py:196-205); only filters_rate=1 is runnable without a middle block, in the reference as well"
            )
        self.middle_slot = cur
        x_in = fmid
        # pools so far: one per encoder block but the very first conv block of the network, plus the final one; a stem adds its own
        # final pool, so with a stem the deepest feature sits at 2 * max_stride (unet.py:178-191)
        stride = 2 ** (self.down_blocks + sb + (1 if sb > 0 else 0))
        self.decoder_stride_to_filters = {stride: x_in}
        skips = skips[::-1]
        if stem_out is not None:
            skips.append(stem_out)  # unet.py:287-288: the stem output is the last decoder block's skip
        for b in range(self.up_blocks):
            fout = int(f * (r ** max(0, self.down_blocks + sb - 1 - b)))
            nxt = stride // 2
            pfx = f"backbone.decoders.0.decoder_stack.{b}.blocks.stack0_dec{b}_s{stride}_to_s{nxt}"
            dst = self._new_slot()
            if self.up_interpolate:
                self.ops.append(OpSpec(L.OP_UPSAMPLE, cur, -1, dst, cur_c, label=pfx + "_interp_bilinear"))
                up_c = cur_c
            else:
                name = pfx + "_trans_conv"
                self.param_shapes[name + ".weight"] = (cur_c, fout, 3, 3)
                self.param_shapes[name + ".bias"] = (fout,)
                self.ops.append(OpSpec(L.OP_CONVT, cur, -1, dst, cur_c, 0, fout, 3, L.FLAG_RELU, name + ".weight", name + ".bias", label=name))
                self.labels[name] = dst
                up_c = fout
            cur, cur_c = dst, up_c
            for i in range(self.convs_per_block):
                name = pfx + f"_refine_conv{i}"
                if i == 0 and b < len(skips):
                    sk, sk_c = skips[b]
                    cur = self._conv(name, sk, sk_c, fout, src1=cur, cin1=cur_c)  # concat (skip, x)
                else:
                    cur = self._conv(name, cur, cur_c, fout)
                cur_c = fout
            self.decoder_stride_to_filters[nxt] = fout
            self.decoder_slot_of_stride[nxt] = cur
            stride = nxt

    @property
    def max_channels(self) -> int:
        return int(self.filters * (self.filters_rate ** (self.down_blocks + self.stem_blocks)))
