"""Backward parity of the cfg3 network at a size with interior tiles (the unit tests use smaller nets): python tools/big_backward_check.py"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from oracle import cpu_ref as O
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.training.module import TrainingModule
bb, heads, mt = bench.CFG3_BB, bench.CFG3_HEADS, "bottomup"
for h in heads.values(): h.setdefault("loss_weight", 1.0)
sd = O.init_state(bb, heads, mt, seed=3, head_scale=1.0)
g = torch.Generator().manual_seed(3)
img = torch.randint(0, 256, (2, 1, 256, 320), dtype=torch.uint8, generator=g)
ref_out = O.model_forward(sd, bb, heads, mt, img)
targets = {k: torch.rand(v.shape, generator=g) * 0.5 for k, v in ref_out.items()}
m = Model("unet", bb, heads, mt); m.load_state_dict(sd)
lw = [h.loss_weight for h in m.heads]
tm = TrainingModule(m, "cuda:0", loss_weights=lw)
ref_losses, ref_grads = O.training_step(sd, bb, heads, mt, img, targets, lw)
loss = tm.forward_backward(img, targets).cpu().numpy()
got = tm.named_grads()
errs = sorted(((float((got[k] - r).abs().max()) / max(float(r.abs().max()), 1e-12), k) for k, r in ref_grads.items()), reverse=True)
print("cfg3 network 256x320 B=2: worst grad errs", [(f"{e:.2e}", k) for e, k in errs[:4]], "loss", loss, ref_losses)
