import sys
sys.path.insert(0, ".")
import torch
import benchlegs.common as Cm
import benchlegs.infer_cfg5 as L5
import benchlegs.single_instance as SI
dev = torch.device("cuda", 0)
for warm in (0.0, 60.0, 0.0, 60.0):
    Cm.WARM_MS = warm
    r = L5.infer_cfg5_leg(50, dev)
    print("WARM_MS", warm, "cfg5 value", round(r["value"]), "forward_ms", r.get("forward_ms"), "mfma_frac", r["roofline"]["frac"] if "roofline" in r else None, flush=True)
