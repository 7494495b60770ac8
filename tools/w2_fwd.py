"""cfg3 forward only, for profilers: python tools/w2_fwd.py [n_forwards] [conv_wino2d 0|1]"""
import sys
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
v = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator().manual_seed(4321)
frames = torch.randint(0, 256, (32, 1, bench.SIZE, bench.SIZE), dtype=torch.uint8, generator=g).to("cuda:0")
m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to("cuda:0").set_option("conv_wino2d", v)
for _ in range(n):
    m(frames)
torch.cuda.synchronize()
print("done")
