"""Two-stage top-down inference (centroid -> crops -> centered instance) on the reference's fixture models (tests/golden/topdown.npz): wall time per batch,
host profile and kernel-side per-call breakdown.  python tools/topdown_profile.py [batch]"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tests import _golden as G
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.backends import HipBackend
from sleap_nn_amd.inference.layers import CenteredInstanceLayer, CentroidLayer, PostprocessConfig, TopDownLayer

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
z = G.load("topdown.npz")
cfg = G.config(z)
cc, ci = cfg["centroid"], cfg["centered"]
wz = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
mc = Model("unet", cc["backbone"], cc["heads"], "centroid"); mc.load_state_dict(wz("wc/"))
mi = Model("unet", ci["backbone"], ci["heads"], "centered_instance"); mi.load_state_dict(wz("wi/"))
pc = PostprocessConfig(peak_threshold=0.03, max_instances=6)
cl = CentroidLayer(HipBackend(mc, DEV), cc["heads"]["confmaps"]["output_stride"], max_instances=6, max_stride=cc["backbone"]["max_stride"], postprocess_config=pc)
il = CenteredInstanceLayer(HipBackend(mi, DEV), ci["heads"]["confmaps"]["output_stride"], max_stride=ci["backbone"]["max_stride"], postprocess_config=PostprocessConfig(peak_threshold=0.03))
td = TopDownLayer(cl, il, (cfg["crop_size"], cfg["crop_size"]))
img = torch.from_numpy(z["image"])
img = img.repeat((B + img.shape[0] - 1) // img.shape[0], 1, 1, 1, 1)[:B].to(DEV) if img.dim() == 5 else img.repeat((B + img.shape[0] - 1) // img.shape[0], 1, 1, 1)[:B].to(DEV)
print("frames", tuple(img.shape), "crop", cfg["crop_size"])
for _ in range(5):
    out = td.predict(img)
torch.cuda.synchronize()
n = 50
t = time.perf_counter()
for _ in range(n):
    out = td.predict(img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
print(f"top-down B={B}: {dt*1e3:.3f} ms / batch = {B/dt:.0f} frames/s")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    out = td.predict(img)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
