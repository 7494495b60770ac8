"""``Model`` = backbone + heads, executed by libposehip on an MI355X.

Drop-in for ``sleap_nn.architectures.model.Model`` (architectures/model.py:157-261): same
constructor signature, same ``state_dict`` key names (so reference checkpoints load), same
``forward(x) -> {head class name: (B, C, H/s, W/s) fp32 NCHW tensor}``.  There are no torch
modules underneath: weights are re-packed once and the forward is one C-ABI call that
enqueues hand-written gfx950 kernels on the current torch stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.heads import ClassMapsHead, ClassVectorsHead, Head, get_head
from sleap_nn_amd.architectures.convnext import ConvNextWrapper
from sleap_nn_amd.architectures.unet import OpSpec, UNet
from sleap_nn_amd.utils import cfg_get, cfg_keys


PRECISIONS = {"exact": 0, "split": 1, "fp16": 2}


def get_backbone(backbone: str, backbone_config):
    """architectures/model.py:36-67.  ``unet`` and ``convnext`` are built natively."""
    if backbone == "unet":
        return UNet.from_config(backbone_config)
    if backbone == "convnext":
        return ConvNextWrapper.from_config(backbone_config)
    if backbone in ("swint", "pretrained"):
        raise NotImplementedError(f"backbone '{backbone}' is not implemented on the MI355X path yet")
    raise KeyError(f"Unsupported backbone: {backbone}. Supported backbones are: unet, convnext")


class Model:
    """Backbone + 1x1 heads (architectures/model.py:157-261)."""

    def __init__(self, backbone_type: str, backbone_config, head_configs, model_type: str) -> None:
        self.backbone_type = backbone_type
        self.backbone_config = backbone_config
        self.head_configs = head_configs
        self.model_type = model_type
        self.heads: List[Head] = get_head(model_type, head_configs)
        self.backbone: UNet = get_backbone(backbone_type, backbone_config)
        self.in_channels = int(cfg_get(backbone_config, "in_channels", 1))

        self.ops: List[OpSpec] = list(self.backbone.ops)
        self.param_shapes = dict(self.backbone.param_shapes)
        self._head_ops_start = len(self.ops)
        for i, head in enumerate(self.heads):
            if isinstance(head, ClassVectorsHead):
                self._add_class_vector_head(i, head)
                continue
            s2f = self.backbone.decoder_stride_to_filters
            if head.output_stride not in self.backbone.decoder_slot_of_stride:
                if not self.backbone.decoder_slot_of_stride and head.output_stride in s2f:
                    src = self.backbone.middle_slot
                else:
                    raise ValueError(
                        f"Head '{head.name}' needs a feature map at output_stride {head.output_stride}, "
                        f"backbone produces strides {sorted(self.backbone.decoder_slot_of_stride)}"
                    )
            else:
                src = self.backbone.decoder_slot_of_stride[head.output_stride]
            cin = s2f[head.output_stride]
            name = f"head_layers.{i}.{head.name}.0"
            self.param_shapes[name + ".weight"] = (head.channels, cin, 1, 1)
            self.param_shapes[name + ".bias"] = (head.channels,)
            flags = L.FLAG_SIGMOID if isinstance(head, ClassMapsHead) else 0
            self.ops.append(OpSpec(L.OP_HEAD, src, -1, -1, cin, 0, head.channels, 1, flags, name + ".weight", name + ".bias", out_index=i, label=name))
        self.unfused_ops = list(self.ops)
        self.fused_ops = self._fuse_cnblocks(self._fuse_pools(self._fuse_stem(self.ops)))
        self.ops = self.fused_ops
        self._state: Dict[str, torch.Tensor] = {k: torch.zeros(v, dtype=torch.float32) for k, v in self.param_shapes.items()}
        self._handle = None
        self._handle_device: Optional[torch.device] = None
        self._workspace: Optional[torch.Tensor] = None
        self.device = torch.device("cpu")
        # bumped whenever the native handle or the workspace is replaced: anything that captured raw pointers into them
        # (HipBackend's hipGraphs) compares generations before replaying
        self.generation = 0
        self._options: Dict[str, float] = {}
        self.precision = "exact"
        self.keep_activations = False
        # a TrainingModule registers its device parameter arena here; after any recompile the packed weights are
        # re-gathered from it, so eval()/train() round trips never fall back to the host copy in _state
        self._live_params: Optional[torch.Tensor] = None
        self._options["workspace_reuse"] = 1.0  # the model starts in the inference (fused) program

    def _add_class_vector_head(self, i: int, head: "ClassVectorsHead") -> None:
        """heads.py:506-539 on the decoder's input feature (architectures/model.py:197-199,253-255)."""
        if not head.global_pool:
            raise ValueError("ClassVectorsHead without global_pool is not supported (the reference flattens a fixed-size feature map)")
        bbn = self.backbone
        src = bbn.middle_slot
        cin = int(bbn.max_channels)
        dst = bbn._new_slot()
        self.ops.append(OpSpec(L.OP_GLOBAL_MAXPOOL, src, -1, dst, cin, label=f"head_layers.{i}.pre_classification_global_pool"))
        cur, cur_c = dst, cin
        for j in range(head.num_fc_layers):
            name = f"head_layers.{i}.pre_classification{j}_fc"
            self.param_shapes[name + ".weight"] = (head.num_fc_units, cur_c)
            self.param_shapes[name + ".bias"] = (head.num_fc_units,)
            dst = bbn._new_slot()
            self.ops.append(OpSpec(L.OP_LINEAR, cur, -1, dst, cur_c, 0, head.num_fc_units, 1, L.FLAG_RELU, name + ".weight", name + ".bias", label=name))
            cur, cur_c = dst, head.num_fc_units
        name = f"head_layers.{i}.{head.name}"
        self.param_shapes[name + ".weight"] = (head.channels, cur_c)
        self.param_shapes[name + ".bias"] = (head.channels,)
        self.ops.append(OpSpec(L.OP_HEAD, cur, -1, -1, cur_c, 0, head.channels, 1, L.FLAG_SOFTMAX, name + ".weight", name + ".bias", out_index=i, label=name))

    @staticmethod
    def _fuse_stem(ops: List[OpSpec]) -> List[OpSpec]:
        """Plan-level fusion of the first encoder block: INPUT_CONV -> CONV -> POOL becomes one
        PH_OP_STEM launch when both convs have <= 16 filters (the full-resolution activations then
        never round-trip through HBM; the full-res output is still written if a decoder block or a
        head reads it)."""
        if len(ops) < 3:
            return ops
        a, b, c = ops[0], ops[1], ops[2]
        ok = (
            a.kind == L.OP_INPUT_CONV and b.kind == L.OP_CONV and c.kind == L.OP_POOL and b.src0 == a.dst and b.src1 < 0
            and c.src0 == b.dst and a.cout <= 16 and b.cout <= 16 and a.cin0 in (1, 3) and a.ksize == 3 and b.ksize == 3
            and not any(o.src0 == a.dst or o.src1 == a.dst for o in ops[2:])
        )
        if not ok:
            return ops
        full_needed = any(o.src0 == b.dst or o.src1 == b.dst for o in ops[3:])
        fused = OpSpec(
            L.OP_STEM, -1, -1, b.dst if full_needed else -1, a.cin0, 0, b.cout, 3, L.FLAG_RELU, a.weight, a.bias,
            label=b.label + "+pool(fused stem)", dst2=c.dst, weight2=b.weight, bias2=b.bias, cmid=a.cout,
        )
        return [fused] + ops[3:]

    @staticmethod
    def _fuse_pools(ops: List[OpSpec]) -> List[OpSpec]:
        """CONV(+ReLU) followed by the POOL of its output -> the conv's epilogue also writes the
        pooled tensor (``dst2``); the full-resolution output is still written (it is the skip)."""
        import copy

        out: List[OpSpec] = []
        i = 0
        while i < len(ops):
            op = ops[i]
            nxt = ops[i + 1] if i + 1 < len(ops) else None
            if op.kind == L.OP_CONV and (op.flags & L.FLAG_RELU) and op.dst2 < 0 and nxt is not None and nxt.kind == L.OP_POOL and nxt.src0 == op.dst:
                f = copy.copy(op)
                f.dst2 = nxt.dst
                f.label = op.label + "+pool"
                out.append(f)
                i += 2
                continue
            out.append(op)
            i += 1
        return out

    @staticmethod
    def _fuse_cnblocks(ops: List[OpSpec]) -> List[OpSpec]:
        """LINEAR -> GELU and LINEAR -> SCALE_ADD pairs become one row GEMM with the activation /
        ``layer_scale * u + x`` in its epilogue (the 4C-wide pre-activation and the un-scaled block
        output then never exist in HBM)."""
        import copy

        def readers(slot, skip):
            return [o for o in ops if o is not skip and (o.src0 == slot or o.src1 == slot)]

        out: List[OpSpec] = []
        i = 0
        while i < len(ops):
            op = ops[i]
            nxt = ops[i + 1] if i + 1 < len(ops) else None
            if op.kind == L.OP_LINEAR and op.flags == 0 and nxt is not None and nxt.src0 == op.dst and not readers(op.dst, nxt):
                if nxt.kind == L.OP_GELU:
                    f = copy.copy(op)
                    f.flags, f.dst = L.FLAG_GELU, nxt.dst
                    out.append(f)
                    i += 2
                    continue
                if nxt.kind == L.OP_SCALE_ADD:
                    f = copy.copy(op)
                    f.flags, f.dst, f.src1, f.weight2, f.label = L.FLAG_SCALE_RESIDUAL, nxt.dst, nxt.src1, nxt.weight, nxt.label
                    out.append(f)
                    i += 2
                    continue
            out.append(op)
            i += 1
        return out

    @classmethod
    def from_config(cls, backbone_type, backbone_config, head_configs, model_type) -> "Model":
        return cls(backbone_type, backbone_config, head_configs, model_type)

    # -- parameters -----------------------------------------------------------------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._state)

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Accepts ``Model`` keys or LightningModule keys (``model.`` prefix, loaders.py:144-176)."""
        sd = {}
        for k, v in state_dict.items():
            k = k[len("model.") :] if k.startswith("model.") else k
            sd[k] = v
        missing = [k for k in self.param_shapes if k not in sd]
        unexpected = [k for k in sd if k not in self.param_shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        for k, shape in self.param_shapes.items():
            if k in sd:
                t = torch.as_tensor(sd[k]).detach().to("cpu", torch.float32).contiguous()
                if tuple(t.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {tuple(shape)}")
                self._state[k] = t
        if self._live_params is not None:  # a TrainingModule owns the parameters on the device: keep its arena in step
            self._live_params.copy_(self.flat_params())
        self._release()
        return missing, unexpected

    def init_xavier_(self, seed: int = 0, head_scale: float = 1.0) -> "Model":
        """Fresh weights the way the reference's trainer initialises a model (``xavier_init_weights``,
        training/utils.py:72-78): Xavier-uniform Conv2d / Linear weights, zero biases; LayerNorm affine = (1, 0),
        ConvNeXt ``layer_scale`` = 1e-6 (the module defaults).  ``head_scale`` shrinks the head weights (synthetic
        benchmarks keep the raw outputs O(1) that way).  Deterministic in ``seed``."""
        import math

        g = torch.Generator().manual_seed(int(seed))
        sd = {}
        for k, shape in self.param_shapes.items():
            if k.endswith(".layer_scale"):
                sd[k] = torch.full(shape, 1e-6)
            elif len(shape) == 1:
                ln = k.endswith(".weight") and any(o.kind == L.OP_LAYERNORM and o.weight == k for o in self.unfused_ops)
                sd[k] = torch.ones(shape) if ln else torch.zeros(shape)
            else:
                rf = 1
                for d in shape[2:]:
                    rf *= int(d)
                fan_in, fan_out = int(shape[1]) * rf, int(shape[0]) * rf
                bound = math.sqrt(6.0 / (fan_in + fan_out))
                w = (torch.rand(shape, generator=g) * 2 - 1) * bound
                sd[k] = w * head_scale if k.startswith("head_layers.") else w
        self.load_state_dict(sd, strict=True)
        return self

    def num_parameters(self) -> int:
        return sum(int(torch.tensor(s).prod()) for s in self.param_shapes.values())

    def eval(self) -> "Model":
        return self.set_fusion(True)

    def train(self, mode: bool = True) -> "Model":
        """Training keeps every activation: the op-by-op (unfused) program runs."""
        return self.set_fusion(not mode)

    def set_fusion(self, fused: bool) -> "Model":
        want = self.fused_ops if fused else self.unfused_ops
        if want is not self.ops:
            self.ops = want
            self._release()
        # the training program keeps fp32 activations for the backward pass; inference runs at `self.precision` and lets
        # activation slots share memory once their last reader has run (unless somebody wants to read them back)
        self.set_option("conv_precision", PRECISIONS[self.precision] if fused else 0)
        self.set_option("workspace_reuse", 1 if (fused and not self.keep_activations) else 0)
        return self

    def set_keep_activations(self, keep: bool) -> "Model":
        """``True``: every activation of an inference forward stays readable (``read_activation``) at the price of one memory
        range per slot (9.0 instead of 3.6 GB at cfg3 x 32 frames)."""
        self.keep_activations = bool(keep)
        if self.ops is self.fused_ops:
            self.set_option("workspace_reuse", 0 if self.keep_activations else 1)
        return self

    def set_precision(self, precision: str) -> "Model":
        """Arithmetic of the inference forward's 3x3 convolutions (UNet programs; ConvNeXt programs always run exact):
        ``"exact"``: fp32 products on the fp32 matrix pipe (bit-for-bit a k-ordered fmaf chain);
        ``"split"``: every operand as a (hi, lo) pair of fp16 numbers, three fp16 MFMAs per product with fp32 accumulation --
        22-bit products, ~1e-6 relative on the head outputs, 16/3 of the fp32 matrix rate;
        ``"fp16"``: plain fp16 operands and storage, fp32 accumulation -- the reference's autocast mode
        (torch_backend.py:113-143; its tolerance is 5e-3)."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
        self.precision = precision
        if self.ops is self.fused_ops:
            self.set_option("conv_precision", PRECISIONS[precision])
        return self

    def param_keys(self) -> List[str]:
        """Order of the canonical flat parameter arena (= ph_model_create's weights[])."""
        return list(self.param_shapes.keys())

    def flat_params(self) -> torch.Tensor:
        return torch.cat([self._state[k].reshape(-1) for k in self.param_keys()])

    def load_flat_params(self, flat: torch.Tensor) -> None:
        """Host copy of the arena back into the state dict (e.g. for checkpointing)."""
        flat = flat.detach().to("cpu", torch.float32)
        o = 0
        for k in self.param_keys():
            n = self._state[k].numel()
            self._state[k] = flat[o : o + n].reshape(self.param_shapes[k]).clone()
            o += n

    def to(self, device) -> "Model":
        self.device = torch.device(device)
        return self

    # -- native handle ----------------------------------------------------------------------
    def _release(self) -> None:
        if self._handle is not None:
            L.lib().ph_model_destroy(self._handle)
            self._handle = None
            self.generation += 1

    def set_option(self, key: str, value: float) -> "Model":
        """Per-handle kernel-variant option (``ph_model_set_option``; keys in include/posehip.h).  Remembered across recompiles."""
        if self._handle is not None:
            L.check(L.lib().ph_model_set_option(self._handle, str(key).encode(), float(value)))
        if self._options.get(str(key)) != float(value):
            self.generation += 1  # a captured hipGraph replays the kernels it was captured with: stale after a variant change
        self._options[str(key)] = float(value)
        return self

    def get_option(self, key: str) -> float:
        if self._handle is None:
            raise RuntimeError("get_option needs a compiled handle (run a forward first)")
        v = C.c_double()
        L.check(L.lib().ph_model_get_option(self._handle, str(key).encode(), C.byref(v)))
        return v.value

    def bind_live_params(self, flat_dev: Optional[torch.Tensor]) -> None:
        """Device arena (canonical order) that owns the live parameters, or None to go back to the host state dict."""
        self._live_params = flat_dev

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _compile(self, device: torch.device) -> None:
        lib = L.lib()
        keys = list(self.param_shapes.keys())
        index = {k: i for i, k in enumerate(keys)}
        descs = (L.OpDesc * len(self.ops))()
        for i, op in enumerate(self.ops):
            d = descs[i]
            d.kind, d.src0, d.src1, d.dst = op.kind, op.src0, op.src1, op.dst
            d.cin0, d.cin1, d.cout, d.ksize, d.flags = op.cin0, op.cin1, op.cout, op.ksize, op.flags
            d.weight = index[op.weight] if op.weight else -1
            d.bias = index[op.bias] if op.bias else -1
            d.out_index = op.out_index
            d.dst2, d.cmid = op.dst2, op.cmid
            d.weight2 = index[op.weight2] if op.weight2 else -1
            d.bias2 = index[op.bias2] if op.bias2 else -1
        tensors = [self._state[k].contiguous() for k in keys]
        ptrs = (C.c_void_p * len(keys))(*[t.data_ptr() for t in tensors])
        numel = (C.c_int64 * len(keys))(*[t.numel() for t in tensors])
        with torch.cuda.device(device):
            h = lib.ph_model_create(descs, len(self.ops), ptrs, numel, len(keys), self.backbone.n_slots, len(self.heads))
        if not h:
            raise L.PosehipError(L.PH_E_INVALID, lib.ph_last_error().decode())
        self._handle = C.c_void_p(h)
        self._handle_device = device
        self.generation += 1
        for k, v in self._options.items():
            L.check(lib.ph_model_set_option(self._handle, k.encode(), v))
        if self._live_params is not None:  # the host copy in _state may be stale during training
            if self._live_params.device != device or self._live_params.numel() != sum(t.numel() for t in tensors):
                raise RuntimeError("live parameter arena does not match this model / device")
            with torch.cuda.device(device):
                L.check(lib.ph_model_set_params(self._handle, C.c_void_p(self._live_params.data_ptr()), L.current_stream_ptr()))

    def _ensure(self, device: torch.device) -> None:
        if self._handle is None or self._handle_device != device:
            self._release()
            self._compile(device)

    def output_shapes(self, height: int, width: int):
        lib = L.lib()
        out = []
        for i in range(len(self.heads)):
            c, h, w = C.c_int32(), C.c_int32(), C.c_int32()
            L.check(lib.ph_model_output_shape(self._handle, i, height, width, C.byref(c), C.byref(h), C.byref(w)))
            out.append((c.value, h.value, w.value))
        return out

    # -- forward --------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, in_dtype: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """``x``: (B, C, H, W) on the GPU.  uint8 is divided by 255 inside the first kernel;
        float input is taken as already normalised unless ``in_dtype == 2`` (0..255 floats)."""
        if x.dim() != 4:
            raise ValueError(f"expected (B, C, H, W), got {tuple(x.shape)}")
        if not x.is_cuda:
            dev = self.device if self.device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
            x = x.to(dev, non_blocking=True)
        device = x.device
        if x.dtype == torch.uint8:
            code = 0
        else:
            x = x.to(torch.float32)
            code = 1 if in_dtype is None else in_dtype
        if x.shape[1] != self.in_channels:
            # architectures/model.py:239-245 (gray <-> rgb); rare path, done with torch plumbing
            xf = x.float() / 255.0 if code in (0, 2) else x
            if x.shape[1] == 1:
                xf = xf.repeat(1, 3, 1, 1)
            elif x.shape[1] == 3:
                r, g, b = xf.unbind(dim=1)
                xf = (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(1)
            x, code = xf, 1
        x = x.contiguous()
        B, Cin, H, W = x.shape
        self._ensure(device)
        lib = L.lib()
        with torch.cuda.device(device):
            need = L.check(lib.ph_model_workspace_bytes(self._handle, B, H, W))
            if self._workspace is None or self._workspace.numel() < need or self._workspace.device != device:
                self._workspace = None
                self._workspace = torch.empty(int(need), dtype=torch.uint8, device=device)
                self.generation += 1
            outs = [torch.empty((B, c, h, w), dtype=torch.float32, device=device) for (c, h, w) in self.output_shapes(H, W)]
            optrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
            L.check(
                lib.ph_model_forward(
                    self._handle, C.c_void_p(x.data_ptr()), code, B, Cin, H, W, C.c_void_p(self._workspace.data_ptr()),
                    self._workspace.numel(), optrs, L.current_stream_ptr(),
                )
            )
        return {head.name: (o.flatten(1) if isinstance(head, ClassVectorsHead) else o) for head, o in zip(self.heads, outs)}

    __call__ = forward

    # -- measurement support ------------------------------------------------------------
    def set_profiling(self, enabled, resume: bool = False) -> None:
        """Per-op HIP-event timing on the forward's stream (bench.py roofline).  ``resume=True`` switches recording back on
        without clearing what was accumulated (sampling every n-th forward of a timed region)."""
        L.check(L.lib().ph_model_set_profiling(self._handle, (2 if resume else 1) if enabled else 0))

    def read_profile(self):
        """-> (list of accumulated ms per op, number of forwards covered)."""
        n = len(self.ops)
        arr = (C.c_double * n)()
        k = C.c_int32()
        L.check(L.lib().ph_model_profile_read(self._handle, arr, n, C.byref(k)))
        return list(arr), k.value

    def last_kernels(self):
        """PH_KV_* code (``_lib.KV_*``) of the kernel family each op of the LAST forward ran; the library's own record of
        its dispatch (``ph_model_last_kernels``), so FLOP accounting never restates the dispatch rules."""
        n = len(self.ops)
        arr = (C.c_int32 * n)()
        L.check(L.lib().ph_model_last_kernels(self._handle, arr, n))
        return list(arr)

    def op_table(self, batch: int, height: int, width: int):
        """Per op: label, kind, algorithmic FLOPs (2*Cin*Cout*k*k*Hout*Wout*B for convolutions, SURVEY
        section 8d) and algorithmic HBM bytes (input read once + output written once, logical channels)."""
        hw = {-1: (height, width)}
        ch = {-1: self.in_channels}
        rows = []
        for op in self.ops:
            h, w = hw[op.src0]
            if op.kind == L.OP_STEM:
                f0 = 2.0 * op.cin0 * op.cmid * 9 * h * w * batch
                f1 = 2.0 * op.cmid * op.cout * 9 * h * w * batch
                ph, pw = (h + 1) // 2, (w + 1) // 2
                byt = op.cin0 * h * w * batch + 4 * op.cout * ph * pw * batch + (4 * op.cout * h * w * batch if op.dst >= 0 else 0)
                rows.append({"label": op.label.split(".")[-1], "kind": op.kind, "flops": f0 + f1, "mfma_flops": f1, "bytes": float(byt)})
                if op.dst >= 0:
                    hw[op.dst] = (h, w)
                hw[op.dst2] = (ph, pw)
                continue
            oh, ow = h, w
            if op.kind == L.OP_PATCH_STEM:
                oh, ow = (h + 2 - op.ksize) // op.cmid + 1, (w + 2 - op.ksize) // op.cmid + 1
            elif op.kind == L.OP_PATCH_CONV:
                oh, ow = h // 2, w // 2
            if op.kind == L.OP_POOL:
                oh, ow = (h + 1) // 2, (w + 1) // 2
            elif op.kind in (L.OP_UPSAMPLE, L.OP_CONVT):
                oh, ow = 2 * h, 2 * w
            cin = op.cin0 + (op.cin1 if op.kind not in (L.OP_LINEAR, L.OP_SCALE_ADD) else 0)
            cout = op.cout if op.kind not in (L.OP_POOL, L.OP_UPSAMPLE) else op.cin0
            flops = 0.0
            if op.kind in (L.OP_CONV, L.OP_INPUT_CONV):
                flops = 2.0 * cin * cout * op.ksize * op.ksize * oh * ow * batch
            elif op.kind == L.OP_CONVT:
                flops = 2.0 * cin * cout * 9 * h * w * batch
            elif op.kind in (L.OP_HEAD, L.OP_LINEAR):
                flops = 2.0 * cin * cout * oh * ow * batch
            elif op.kind in (L.OP_PATCH_STEM, L.OP_PATCH_CONV):
                flops = 2.0 * cin * cout * op.ksize * op.ksize * oh * ow * batch
            elif op.kind == L.OP_DWCONV:
                flops = 2.0 * cout * 49 * oh * ow * batch
            in_bytes = (1 if op.src0 < 0 else 4) * op.cin0 * h * w * batch + 4 * op.cin1 * h * w * batch
            out_bytes = 4 * cout * oh * ow * batch
            rows.append({"label": op.label.split(".")[-1], "kind": op.kind, "flops": flops, "bytes": float(in_bytes + out_bytes),
                         "cin0": op.cin0, "cin1": op.cin1, "cout": cout, "ksize": op.ksize, "out_hw": (oh, ow)})
            if op.kind != L.OP_HEAD:
                hw[op.dst] = (oh, ow)
            if op.kind == L.OP_CONV and op.dst2 >= 0:
                hw[op.dst2] = ((oh + 1) // 2, (ow + 1) // 2)
        return rows

    def read_activation(self, conv_name: str, batch: int, height_width) -> torch.Tensor:
        """Debug/parity: NCHW copy of the activation a named conv produced in the last forward."""
        if self.ops is self.fused_ops and not self.keep_activations:
            raise RuntimeError("activation slots are recycled during an inference forward: call set_keep_activations(True) before the forward")
        slot = self.backbone.labels[conv_name]
        op = next((o for o in self.ops if o.dst == slot), None)
        if op is None:
            raise KeyError(f"activation of {conv_name} is fused away (it never exists in HBM)")
        h, w = height_width
        out = torch.empty((batch, op.cout, h, w), dtype=torch.float32, device=self._handle_device)
        L.check(L.lib().ph_model_read_slot(self._handle, slot, C.c_void_p(out.data_ptr()), out.numel(), L.current_stream_ptr()))
        return out
