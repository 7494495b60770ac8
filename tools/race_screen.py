"""Race screen for the pipelined conv kernels (counted vmcnt + barriers): the cfg3 forward and a ConvNeXt forward repeated many
times at several batch sizes must reproduce the first run's bits every time.  python tools/race_screen.py [reps]"""
import sys
sys.path.insert(0, ".")
import torch, bench
from sleap_nn_amd.architectures.model import Model
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(4321)
bad = 0
for B, size in ((32, 1024), (5, 1024), (3, 768), (7, 544), (1, 1024)):
    frames = torch.randint(0, 256, (B, 1, size, size), dtype=torch.uint8, generator=g).to("cuda:0")
    m = Model("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup").init_xavier_(seed=1234, head_scale=0.05).to("cuda:0")
    ref = {k: v.clone() for k, v in m(frames).items()}
    n = max(4, reps * 32 // max(B * size * size // (1024 * 1024), 1) // 8)
    for r in range(n):
        out = m(frames)
        for k in ref:
            if not torch.equal(out[k], ref[k]):
                bad += 1
                print(f"MISMATCH B={B} size={size} rep={r} head={k} max diff {(out[k]-ref[k]).abs().max().item():.3g}", flush=True)
    print(f"B={B} size={size}: {n} repetitions", "ok" if not bad else "BAD", flush=True)
print("mismatches:", bad)
