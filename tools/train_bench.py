"""Training-step timing at cfg3 shapes (not the headline metric): forward + loss + backward + Adam."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.training.module import TrainingModule

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup")
m.init_xavier_(seed=1234, head_scale=0.05)
tm = TrainingModule(m, "cuda:0", lr=1e-4)
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, generator=g).cuda()
tg = {"MultiInstanceConfmapsHead": torch.rand((B, 13, S // 4, S // 4), generator=g).cuda(), "PartAffinityFieldsHead": torch.rand((B, 24, S // 8, S // 8), generator=g).cuda()}
for _ in range(2):
    loss = tm.training_step({"image": img, **tg})
torch.cuda.synchronize()
t = time.perf_counter()
n = 5
for _ in range(n):
    loss = tm.training_step({"image": img, **tg})
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
fl = 3 * sum(r["flops"] for r in m.op_table(1, S, S)) * B
print(f"train step B={B} {S}x{S}: {dt*1e3:.1f} ms = {B/dt:.1f} frames/s; ~{fl/dt/1e12:.1f} TFLOP/s (3x fwd conv FLOPs); loss {loss.cpu().numpy()}")
from sleap_nn_amd import _lib as L
names = bench.conv_kernel_short_names()
kv = m.last_kernels()
tab = m.op_table(B, S, S)
print("forward kernels of the training plan:", [(r["label"].replace("stack0_", ""), names.get(c, c)) for r, c in zip(tab, kv) if r["kind"] == L.OP_CONV])
