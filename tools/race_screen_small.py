"""Race / determinism screen for the small-batch routings (split K on both Winograd kernels, fused heads, the Cout-32 route, one-pass peaks): repeated forwards and peak
finding at small batches must reproduce the first run's bits every time.  python tools/race_screen_small.py [reps]"""
import sys
sys.path.insert(0, ".")
import torch
import bench
from sleap_nn_amd.architectures.model import Model
from sleap_nn_amd.inference.ops.peaks import find_local_peaks_device

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(4321)
bad = 0
cases = [("unet", bench.CFG3_BB, bench.CFG3_HEADS, "bottomup", B, S) for B, S in ((1, 1024), (2, 1024), (4, 1024), (8, 512), (3, 256), (1, 256))]
heads1 = {"confmaps": {"part_names": [str(i) for i in range(5)], "output_stride": 2}}
cases += [("unet", bench.SI_BB, heads1, "single_instance", B, S) for B, S in ((1, 256), (4, 320), (8, 512), (2, 128))]
for kind, bb, heads, mt, B, S in cases:
    frames = torch.randint(0, 256, (B, 1, S, S), dtype=torch.uint8, generator=g).to("cuda:0")
    m = Model(kind, bb, heads, mt).init_xavier_(seed=1234, head_scale=0.05).to("cuda:0")
    ref = {k: v.clone() for k, v in m(frames).items()}
    n = max(8, reps * 4 // max(B * S * S // (256 * 256), 1))
    for r in range(n):
        out = m(frames)
        for k in ref:
            if not torch.equal(out[k], ref[k]):
                bad += 1
                print(f"MISMATCH {mt} B={B} size={S} rep={r} head={k}", flush=True)
    print(f"{mt} B={B} size={S}: {n} repetitions", "ok" if not bad else "BAD", flush=True)
cms, _ = bench.rendered_heads(4, torch.device("cuda", 0))
base = [t.clone() for t in find_local_peaks_device(cms, 0.2, "integral", 5, 4096)[:5]]
for r in range(reps):
    cur = find_local_peaks_device(cms, 0.2, "integral", 5, 4096)[:5]
    n = int(base[4][0])
    for a, b in zip(base[:4], cur[:4]):
        if not torch.equal(a[:n], b[:n]):
            bad += 1
            print("MISMATCH peaks rep", r, flush=True)
print("peaks:", reps, "repetitions")
print("mismatches:", bad)
