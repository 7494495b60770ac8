"""Learning-rate schedules of the reference's ``configure_optimizers`` (training/lightning_modules.py:765-857), as host-side
scalar logic for ``TrainingModule`` (its ``lr`` attribute is read at every optimizer step):

* ``cosine_annealing_warmup`` / ``linear_warmup_linear_decay``: the reference's own closed forms (training/schedulers.py:11-200);
* ``step_lr``: ``torch.optim.lr_scheduler.StepLR`` (lr * gamma ** (epoch // step_size));
* ``reduce_lr_on_plateau``: ``torch.optim.lr_scheduler.ReduceLROnPlateau(mode="min")`` on the validation loss.

Priority when several are configured: cosine > linear > step > plateau (config/trainer_config.py:228-247); with no scheduler the
learning rate stays constant.  One ``step(val_loss)`` per epoch, like Lightning calls the scheduler.
"""
from __future__ import annotations

import math
from typing import Optional


class LRSchedule:
    def __init__(self, base_lr: float, lr_scheduler: Optional[dict] = None, max_epochs: Optional[int] = None) -> None:
        self.base_lr = float(base_lr)
        self.lr = float(base_lr)
        self.epoch = 0
        cfg = dict(lr_scheduler or {})
        self.kind, self.cfg = None, {}
        for name in ("cosine_annealing_warmup", "linear_warmup_linear_decay", "step_lr", "reduce_lr_on_plateau"):
            if cfg.get(name) is not None:
                self.kind, self.cfg = name, dict(cfg[name])
                break
        if self.kind in ("cosine_annealing_warmup", "linear_warmup_linear_decay"):
            w = int(self.cfg.get("warmup_epochs", 5))
            m = self.cfg.get("max_epochs")
            m = int(max_epochs if m is None else m) if (m is not None or max_epochs is not None) else None
            if m is None:
                raise ValueError(f"{self.kind} needs max_epochs (in its config or from the trainer)")
            if w < 0 or m <= 0 or w >= m:
                raise ValueError(f"warmup_epochs ({w}) must be >= 0 and < max_epochs ({m})")
            self.cfg["warmup_epochs"], self.cfg["max_epochs"] = w, m
            self.lr = self._closed_form(0)
        elif self.kind == "reduce_lr_on_plateau":
            self.best = math.inf
            self.num_bad = 0
            self.cooldown_counter = 0

    def _closed_form(self, epoch: int) -> float:
        c = self.cfg
        w, m = c["warmup_epochs"], c["max_epochs"]
        start = float(c.get("warmup_start_lr", 0.0))
        if epoch < w:
            return start + (epoch / w) * (self.base_lr - start)
        progress = min(1.0, (epoch - w) / (m - w))
        if self.kind == "cosine_annealing_warmup":
            eta_min = float(c.get("eta_min", 0.0))
            return eta_min + (self.base_lr - eta_min) * (1 + math.cos(math.pi * progress)) / 2
        end_lr = float(c.get("end_lr", 0.0))
        return self.base_lr + progress * (end_lr - self.base_lr)

    def step(self, val_loss: Optional[float] = None) -> float:
        """End of an epoch: returns (and stores in ``self.lr``) the learning rate of the next one."""
        self.epoch += 1
        if self.kind in ("cosine_annealing_warmup", "linear_warmup_linear_decay"):
            self.lr = self._closed_form(self.epoch)
        elif self.kind == "step_lr":
            self.lr = self.base_lr * float(self.cfg.get("gamma", 0.1)) ** (self.epoch // int(self.cfg.get("step_size", 10)))
        elif self.kind == "reduce_lr_on_plateau":
            if val_loss is None:
                raise ValueError("reduce_lr_on_plateau monitors val/loss: pass it to step()")
            c = self.cfg
            thr, mode = float(c.get("threshold", 1e-6)), c.get("threshold_mode", "abs")
            better = val_loss < (self.best * (1.0 - thr) if mode == "rel" else self.best - thr)
            if better:
                self.best, self.num_bad = float(val_loss), 0
            else:
                self.num_bad += 1
            if self.cooldown_counter > 0:
                self.cooldown_counter -= 1
                self.num_bad = 0
            if self.num_bad > int(c.get("patience", 5)):
                min_lr = c.get("min_lr", 1e-8)
                min_lr = float(min_lr[0] if isinstance(min_lr, (list, tuple)) else min_lr)
                new_lr = max(self.lr * float(c.get("factor", 0.5)), min_lr)
                if self.lr - new_lr > 1e-8:  # torch's eps
                    self.lr = new_lr
                self.cooldown_counter = int(c.get("cooldown", 3))
                self.num_bad = 0
        return self.lr
