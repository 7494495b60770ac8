"""Training-step timing of the ConvNeXt-tiny centered-instance network (cfg4: 384x384 crops, os=2).

    python tools/convnext_train_bench.py [batch] [size]
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.training.module import TrainingModule

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 384
bb = {"model_type": "tiny", "arch": None, "in_channels": 1, "kernel_size": 3, "filters_rate": 2, "convs_per_block": 2, "up_interpolate": True,
      "stem_patch_kernel": 4, "stem_patch_stride": 2, "output_stride": 2, "max_stride": 32}
heads = {"confmaps": {"part_names": [str(i) for i in range(13)], "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0}}
m = Model("convnext", bb, heads, "centered_instance")
m.init_xavier_(seed=1234, head_scale=0.05)
tm = TrainingModule(m, "cuda:0", loss_weights=[1.0])
img = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, device="cuda:0")
tgt = {"CenteredInstanceConfmapsHead": torch.rand((B, 13, S // 2, S // 2), device="cuda:0")}
batch = {"image": img, **tgt}
for _ in range(2):
    tm.training_step(batch)
torch.cuda.synchronize()
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
N = 3
t = time.time()
for _ in range(N):
    loss = tm.training_step(batch)
torch.cuda.synchronize()
dt = (time.time() - t) / N
flops = 3 * 211.6e9 * B * (S / 384) ** 2
print(f"step {dt*1e3:.1f} ms  {B/dt:.1f} crops/s  ~{flops/dt/1e12:.1f} TFLOP/s (3x forward FLOPs)  loss {float(loss[0]):.5f}")
