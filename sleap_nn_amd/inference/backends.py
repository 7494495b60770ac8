"""Runtime backends: the reference's ``ModelBackend`` protocol and its MI355X implementation.

``ModelBackend`` restates the 4-member protocol of
``sleap_nn/inference/layers/backends/base.py:18-79``; ``HipBackend`` is the counterpart of
``TorchBackend`` (``torch_backend.py:46-266``): it owns the model, moves the batch to the
device, resolves the LightningModule-forward preamble (squeeze the n_samples axis,
uint8 -> /255, ``normalize_on_gpu``'s ``max() > 1`` test for float input --
training/lightning_modules.py:1840-1848, data/normalization.py:7-35) and returns a dict of
fp32 NCHW tensors keyed by head class name.
"""
from __future__ import annotations

from typing import Dict, Protocol, Tuple, runtime_checkable

import torch

from sleap_nn_amd import _lib as L


@runtime_checkable
class ModelBackend(Protocol):
    @property
    def device(self) -> str: ...

    @property
    def does_baked_postproc(self) -> bool: ...

    def __call__(self, x: torch.Tensor) -> Dict[str, torch.Tensor]: ...

    def warmup(self, input_shape: Tuple[int, ...]) -> None: ...


class HipBackend:
    """Hand-written-HIP forward behind the ``ModelBackend`` protocol."""

    def __init__(self, model, device: str = "cuda", use_graph: bool = False, precision: str = None, use_fp16: bool = False) -> None:
        """``use_graph``: replay the forward of each (shape, dtype) as ONE hipGraph launch (captured on
        first use through ``torch.cuda.CUDAGraph``): ~25 kernel launches become one, which is what bounds
        small batches (a 256x256 frame is ~0.3 ms of kernels).  The returned tensors are the graph's static
        outputs -- valid until the next call with the same shape."""
        L.lib()  # fail loudly right here if the native library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("HipBackend needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ValueError(f"HipBackend runs on the GPU only, got device={device!r}")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self._device = dev
        # ``use_fp16`` mirrors TorchBackend(use_fp16=True) (torch_backend.py:113-143: autocast); ``precision`` selects among
        # "exact" / "split" / "fp16" explicitly (Model.set_precision); None keeps what the model is set to
        if use_fp16:
            if precision not in (None, "fp16"):
                raise ValueError("use_fp16=True contradicts precision=%r" % (precision,))
            precision = "fp16"
        if precision is not None:
            model.set_precision(precision)
        self.model = model.to(dev).eval()
        self.use_graph = bool(use_graph)
        self._graphs: Dict[tuple, tuple] = {}
        self._graph_generation = -1

    @property
    def device(self) -> str:
        return str(self._device)

    @property
    def does_baked_postproc(self) -> bool:
        return False

    def __call__(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        if not isinstance(x, torch.Tensor):
            raise TypeError(f"backend input must be a torch.Tensor, got {type(x).__name__}")
        x = x.to(self._device, non_blocking=True)
        if x.dim() == 5:
            x = x.squeeze(1)
        x, code = self.input_code(x)
        out = self._forward_graph(x, code) if self.use_graph else self.model.forward(x, in_dtype=code)
        if isinstance(out, torch.Tensor):
            out = {"output": out}
        if not isinstance(out, dict):
            raise TypeError(f"unexpected model output type {type(out).__name__}")
        return out

    @staticmethod
    def input_code(x: torch.Tensor):
        """``(x, in_dtype code)`` for a batch already on the device: uint8 frames keep their dtype (code None: the first kernel divides by 255);
        float frames become fp32 and take ``normalize_on_gpu``'s data-dependent branch (data/normalization.py:32): code 2 (divide by 255) when
        ``max() > 1``, code 1 (as is) otherwise.  One host read of one scalar."""
        if x.dtype == torch.uint8:
            return x, None
        x = x.to(torch.float32)
        return x, (2 if bool(x.max() > 1.0) else 1)

    def _forward_graph(self, x: torch.Tensor, code) -> Dict[str, torch.Tensor]:
        """A captured graph holds raw pointers into the model's workspace and packed weights.  Each entry therefore keeps
        a reference to the workspace tensor it was captured on (so the allocator cannot hand that memory to anybody
        else) and the model generation it saw; any reallocation / recompile / weight reload bumps the generation and
        the stale entries are dropped and re-captured instead of being replayed into freed memory."""
        if self._graph_generation != self.model.generation:
            self._graphs.clear()
        key = (tuple(x.shape), x.dtype, code)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = x.clone()
            stream = torch.cuda.Stream(self._device)
            stream.wait_stream(torch.cuda.current_stream(self._device))
            with torch.cuda.stream(stream):  # warm-up outside the capture: handle creation, workspace allocation
                self.model.forward(static_in, in_dtype=code)
            torch.cuda.current_stream(self._device).wait_stream(stream)
            torch.cuda.synchronize(self._device)
            if self._graph_generation != self.model.generation:  # the warm-up grew the workspace / rebuilt the handle
                self._graphs.clear()
                self._graph_generation = self.model.generation
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.model.forward(static_in, in_dtype=code)
            assert self.model.generation == self._graph_generation, "capture must not reallocate"
            entry = (graph, static_in, static_out, self.model._workspace)
            self._graphs[key] = entry
        graph, static_in, static_out, _ws = entry
        if x.data_ptr() != static_in.data_ptr():  # a caller that fills ``static_input(...)`` itself (e.g. its H2D target) skips this device-to-device copy
            static_in.copy_(x, non_blocking=True)
        graph.replay()
        return static_out

    def static_input(self, shape: Tuple[int, ...], dtype: torch.dtype = torch.uint8) -> torch.Tensor:
        """The captured graph's own input buffer for ``shape`` (4-D ``(B, C, H, W)``; captured on first use).  Writing the frames straight into it -- as the target
        of the H2D copy, say -- and passing it to ``__call__`` makes a step ONE graph launch with no staging copy in front (a 256 x 256 frame: ~5 us of copy + ~9 us of
        launch gap on a ~300-us forward)."""
        if not self.use_graph:
            raise RuntimeError("static_input needs use_graph=True")
        x = torch.zeros(tuple(shape), dtype=dtype, device=self._device)
        code = None if dtype == torch.uint8 else 1
        self._forward_graph(x, code)
        return self._graphs[(tuple(x.shape), x.dtype, code)][1]

    def warmup(self, input_shape: Tuple[int, ...]) -> None:
        x = torch.zeros(tuple(input_shape), dtype=torch.uint8, device=self._device)
        self(x)
        torch.cuda.synchronize(self._device)
