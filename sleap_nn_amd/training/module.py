"""One data-parallel training step on the MI355X: forward, per-head MSE (+OHKM), backward,
gradient all-reduce (RCCL over xGMI through ``torch.distributed``), Adam.

Mirrors the step semantics of the reference's LightningModules
(``sleap_nn/training/lightning_modules.py``: ``training_step`` :1850-1922 for bottom-up,
``_compute_negative_weighted_loss`` :490-545 = ``nn.MSELoss`` per head, loss = sum of
``loss_weight`` x head loss, ``compute_ohkm_loss`` ``training/losses.py:8-63``,
``configure_optimizers`` :750-763 = ``torch.optim.Adam(lr, amsgrad)``) and DDP's gradient averaging.
Parameters, gradients and Adam moments live in flat fp32 arenas in the reference's state_dict
order, so the all-reduce is ONE collective over one contiguous buffer (31 MB at cfg3).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from sleap_nn_amd import _lib as L
from sleap_nn_amd.architectures.model import Model


@dataclass
class OHKMConfig:
    """trainer_config.online_hard_keypoint_mining (losses.py:8-15 defaults)."""

    online_mining: bool = False
    hard_to_easy_ratio: float = 2.0
    min_hard_keypoints: int = 2
    max_hard_keypoints: Optional[int] = None
    loss_scale: float = 5.0


class TrainingModule:
    """Owns the device arenas of one model replica and runs training steps."""

    def __init__(self, model: Model, device: str = "cuda", lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, amsgrad: bool = False,
                 optimizer: str = "Adam", weight_decay: Optional[float] = None,
                 loss_weights: Optional[Sequence[float]] = None, ohkm: Optional[OHKMConfig] = None,
                 negative_loss_weight: float = 1.0, lr_scheduler=None, max_epochs: Optional[int] = None, wino4: bool = True,
                 native_allreduce: Optional[bool] = None) -> None:
        """``negative_loss_weight``: weight of frames flagged ``is_negative`` in the train-stage MSE (lightning_modules.py:149-153,
        526-545).  ``lr_scheduler``: the reference's scheduler config (a name or ``{name: {...}}``, lightning_modules.py:765-857);
        ``self.lr`` then follows it: one ``on_epoch_end(val_loss)`` per epoch, like Lightning steps the scheduler.
        ``wino4``: the K-heavy 3x3 convolutions of the forward and of the data gradients run the Winograd F(4x4,3x3) kernel in the
        training plan too (handle option ``conv_wino4 = 2``; gradients within ~1e-5 of their tensor's scale of the F(2x2,3x3) ones).
        ``native_allreduce``: with more than one rank, exchange the gradients through the C ABI's own RCCL communicator (``ph_model_set_comm``: the two buckets are
        enqueued by ``ph_model_backward`` itself) instead of ``torch.distributed``; default: when the process group's backend is nccl (= RCCL on ROCm)."""
        L.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("TrainingModule needs an MI355X; there is no CPU fallback")
        dev = torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.model = model.train(True).to(dev)
        self.model.set_option("conv_wino4", 2 if wino4 else 1)
        self.lr, self.betas, self.eps, self.amsgrad = lr, betas, eps, amsgrad
        if optimizer not in ("Adam", "AdamW"):
            raise ValueError(f"optimizer must be 'Adam' or 'AdamW' (lightning_modules.py:752-755), got {optimizer!r}")
        # torch's defaults, which is what the reference gets: Adam 0 (L2 form, unused here), AdamW 0.01 (decoupled)
        self.optimizer = optimizer
        self.weight_decay = (0.01 if optimizer == "AdamW" else 0.0) if weight_decay is None else float(weight_decay)
        if optimizer == "Adam" and self.weight_decay != 0.0:
            raise ValueError("Adam with L2 weight decay is not what the reference configures; use optimizer='AdamW'")
        self.loss_weights = [float(w) for w in (loss_weights if loss_weights is not None else [h.loss_weight for h in model.heads])]
        self.ohkm = ohkm or OHKMConfig()
        self.negative_loss_weight = float(negative_loss_weight)
        if isinstance(lr_scheduler, str):  # defaults of the named scheduler (lightning_modules.py:773-785)
            lr_scheduler = {lr_scheduler: {}}
        from sleap_nn_amd.training.schedulers import LRSchedule

        self.schedule = LRSchedule(lr, lr_scheduler, max_epochs) if lr_scheduler else None
        if self.schedule is not None:
            self.lr = self.schedule.lr
        self.params = model.flat_params().to(dev)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.max_exp_avg_sq = torch.zeros_like(self.params) if amsgrad else None
        self.step_count = 0
        self._grad_ws: Optional[torch.Tensor] = None
        self._loss = torch.zeros(1 + len(model.heads), dtype=torch.float32, device=dev)
        # from here on the device arena owns the parameters: every (re)compile of the model's handle -- eval() for a
        # validation pass, train() afterwards, a fusion switch -- re-gathers its packed weights from self.params
        self.model.bind_live_params(self.params)
        self.model._ensure(dev)
        self._bucket_split: Optional[int] = None
        self._bucket_event = None
        self._comm_stream = None
        self._comm = None  # the C ABI's RCCL communicator (parallel.Communicator): ph_model_backward then exchanges the gradients itself
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self._comm_stream = torch.cuda.Stream(dev)
            from sleap_nn_amd.parallel import Communicator

            native = native_allreduce if native_allreduce is not None else (dist.get_backend() == "nccl")
            if native and Communicator.available():
                self._comm = Communicator.create(dev)
            else:  # gloo rehearsals / hosts without librccl: torch.distributed carries the two buckets (parallel.all_reduce_buckets_)
                self._bucket_event = torch.cuda.Event()
                self._bucket_event.record()  # creates the underlying hipEvent_t
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.params, src=0)  # identical initial weights on every rank (DDP semantics)
            self._push_params()

    # ------------------------------------------------------------------------------
    def _push_params(self) -> None:
        with torch.cuda.device(self.device):
            L.check(L.lib().ph_model_set_params(self.model._handle, C.c_void_p(self.params.data_ptr()), L.current_stream_ptr()))

    def forward_backward(self, image: torch.Tensor, targets: Dict[str, torch.Tensor], is_negative: Optional[torch.Tensor] = None,
                         stage: str = "train") -> torch.Tensor:
        """Forward + loss + backward.  ``image``: (B[,1],C,H,W) uint8/float; ``targets``: head name ->
        (B, c, h, w) fp32.  Fills ``self.grads`` (local gradients) and returns the loss tensor
        ``[total, head_0, head_1, ...]`` (device, no sync).  ``is_negative`` (B,) bool: frames without instances; in the
        train stage their MSE is weighted by ``negative_loss_weight`` (``_compute_negative_weighted_loss``), any other
        stage stays unweighted so ``val/loss`` equals plain ``nn.MSELoss``."""
        x = image.to(self.device, non_blocking=True)
        if x.dim() == 5:
            x = x.squeeze(1)
        code = 0
        if x.dtype != torch.uint8:
            x = x.to(torch.float32)
            code = 2 if bool(x.max() > 1.0) else 1
        x = x.contiguous()
        out = self.model.forward(x, in_dtype=None if code == 0 else code)
        B, Cin, H, W = x.shape
        lib = L.lib()
        m = self.model
        with torch.cuda.device(self.device):
            need = L.check(lib.ph_model_backward_workspace_bytes(m._handle, B, H, W))
            if self._grad_ws is None or self._grad_ws.numel() < need:
                self._grad_ws = None
                self._grad_ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            outs = [out[h.name].contiguous() for h in m.heads]
            tg = [targets[h.name].to(self.device, torch.float32).contiguous() for h in m.heads]
            for o, t in zip(outs, tg):
                if tuple(o.shape) != tuple(t.shape):
                    raise ValueError(f"target shape {tuple(t.shape)} != prediction shape {tuple(o.shape)}")
            optr = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
            tptr = (C.c_void_p * len(tg))(*[t.data_ptr() for t in tg])
            lw = (C.c_float * len(self.loss_weights))(*self.loss_weights)
            sw = None
            if is_negative is not None and stage == "train" and self.negative_loss_weight != 1.0:
                neg = torch.as_tensor(is_negative).to(self.device).reshape(-1).bool()
                if neg.numel() != B:
                    raise ValueError(f"is_negative has {neg.numel()} entries for a batch of {B}")
                sw = torch.where(neg, float(self.negative_loss_weight), 1.0).to(torch.float32).contiguous()
            k = self.ohkm
            # (re)bind on every step: the handle may have been recompiled, or last used by another TrainingModule whose event
            # is gone -- a handle never keeps an event this module does not own (NULL = no mid-sweep record)
            if self._bucket_event is not None:
                self._bucket_split = int(L.check(lib.ph_model_grad_bucket_split(m._handle)))
                L.check(lib.ph_model_set_bucket_event(m._handle, C.c_void_p(self._bucket_event.cuda_event)))
            else:
                L.check(lib.ph_model_set_bucket_event(m._handle, None))
            # (the native exchange: bound per step for the same reason; validation passes compute no exchange -- every rank evaluates its own shard)
            if self._comm is not None and stage == "train":
                L.check(lib.ph_model_set_comm(m._handle, C.c_void_p(self._comm.handle), C.c_void_p(self._comm_stream.cuda_stream)))
            else:
                L.check(lib.ph_model_set_comm(m._handle, None, None))
            L.check(
                lib.ph_model_backward(
                    m._handle, C.c_void_p(x.data_ptr()), code, B, Cin, H, W, C.c_void_p(m._workspace.data_ptr()), C.c_void_p(self._grad_ws.data_ptr()),
                    self._grad_ws.numel(), optr, tptr, lw, C.c_void_p(sw.data_ptr()) if sw is not None else None, 1 if k.online_mining else 0, float(k.hard_to_easy_ratio), int(k.min_hard_keypoints),
                    -1 if k.max_hard_keypoints is None else int(k.max_hard_keypoints), float(k.loss_scale), C.c_void_p(self._loss.data_ptr()),
                    C.c_void_p(self.grads.data_ptr()), L.current_stream_ptr(),
                )
            )
        self._last_out = out
        return self._loss

    def all_reduce_grads(self) -> float:
        """Sum the flat gradient arena over the ranks; returns the scale that turns the sum into DDP's mean.

        Two buckets, overlapped with the backward sweep: the arena tail (decoder + heads, final first) is reduced on a side
        stream as soon as the backward has recorded ``_bucket_event`` -- while the encoder's gradients are still being
        computed -- and the head of the arena right after the backward; the compute stream waits for both before Adam.  On
        xGMI a ring all-reduce is bound by one link (~153 GB/s): 31 MB (cfg3 UNet) / 352 MB (ConvNeXt-tiny) take ~0.4 / ~4 ms,
        so one bucket boundary is all the overlap there is to win."""
        if self._comm is not None:  # ph_model_backward has already enqueued both buckets (ph_model_set_comm) and joined the side stream
            return 1.0 / self._comm.world
        from sleap_nn_amd.parallel import all_reduce_buckets_

        return all_reduce_buckets_(self.grads, self._bucket_split, tail_ready=self._bucket_event, comm_stream=self._comm_stream)

    def optimizer_step(self, grad_scale: float = 1.0) -> None:
        self.step_count += 1
        with torch.cuda.device(self.device):
            L.check(
                L.lib().ph_adamw_step(
                    C.c_void_p(self.params.data_ptr()), C.c_void_p(self.grads.data_ptr()), C.c_void_p(self.exp_avg.data_ptr()),
                    C.c_void_p(self.exp_avg_sq.data_ptr()), C.c_void_p(self.max_exp_avg_sq.data_ptr()) if self.max_exp_avg_sq is not None else None,
                    self.params.numel(), float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay), self.step_count,
                    float(grad_scale), L.current_stream_ptr(),
                )
            )
        self._push_params()

    def training_step(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """``batch``: {"image": ..., <head name>: target, ..., ["is_negative": (B,) bool]}.  Returns the loss tensor (device)."""
        targets = {k: v for k, v in batch.items() if k not in ("image", "is_negative")}
        loss = self.forward_backward(batch["image"], targets, batch.get("is_negative"))
        scale = self.all_reduce_grads()
        self.optimizer_step(scale)
        return loss

    def validation_step(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Loss of a validation batch on the current parameters: unweighted (``stage="val"``), no optimizer step.  The
        gradients it leaves in ``self.grads`` are overwritten by the next training step."""
        targets = {k: v for k, v in batch.items() if k not in ("image", "is_negative")}
        return self.forward_backward(batch["image"], targets, batch.get("is_negative"), stage="val").clone()

    def on_epoch_end(self, val_loss: Optional[float] = None) -> float:
        """Step the learning-rate schedule once (Lightning: interval "epoch", monitor "val/loss"); returns the next epoch's lr."""
        if self.schedule is not None:
            self.lr = self.schedule.step(val_loss)
        return self.lr

    def close(self) -> None:
        """Unbind this module's bucket event from the model handle (the handle may outlive the module)."""
        ev, self._bucket_event = self._bucket_event, None
        if ev is not None and getattr(self.model, "_handle", None) is not None:
            L.lib().ph_model_set_bucket_event(self.model._handle, None)
        comm, self._comm = getattr(self, "_comm", None), None
        if comm is not None:
            if getattr(self.model, "_handle", None) is not None:
                L.lib().ph_model_set_comm(self.model._handle, None, None)
            comm.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------
    def named_grads(self) -> Dict[str, torch.Tensor]:
        out, o = {}, 0
        g = self.grads.detach().cpu()
        for k in self.model.param_keys():
            n = int(torch.tensor(self.model.param_shapes[k]).prod())
            out[k] = g[o : o + n].reshape(self.model.param_shapes[k])
            o += n
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        self.model.load_flat_params(self.params)
        return self.model.state_dict()
