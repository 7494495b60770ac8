"""Where the host time of a pipelined batch goes: micro-timings of the staging primitives on this box (development probe)."""
import time

import torch

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
vid = torch.randint(0, 256, (100, 1, 320, 560), dtype=torch.uint8)


def t(label, fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{label:50s} {1e6 * (time.perf_counter() - t0) / n:10.1f} us")


batch = vid[4:8]
t("is_pinned (pageable view)", lambda: batch.is_pinned())
pin = torch.empty(batch.shape, dtype=batch.dtype, pin_memory=True)
t("is_pinned (pinned)", lambda: pin.is_pinned())
t("pinned.copy_(pageable batch)", lambda: pin.copy_(batch))
t("pinned.to(dev, non_blocking)", lambda: pin.to(dev, non_blocking=True))
t("pageable.to(dev, non_blocking)", lambda: batch.to(dev, non_blocking=True))
t("torch.empty(pin_memory=True) 717 KB", lambda: torch.empty(batch.shape, dtype=batch.dtype, pin_memory=True))
t("torch.empty(pin_memory=True) 80 KB", lambda: torch.empty(20000, dtype=torch.float32, pin_memory=True))
ev = torch.cuda.Event()
t("event record + synchronize", lambda: (ev.record(), ev.synchronize()))
d = torch.empty(20000, device=dev)
h = torch.empty(20000, pin_memory=True)
t("D2H 80 KB into pinned, non_blocking", lambda: h.copy_(d, non_blocking=True))
