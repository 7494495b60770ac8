"""Generate golden input/output vectors by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY.  Usage (container only; needs /root/reference)::

    python oracle/gen_golden.py

Writes ``tests/golden/*.npz`` (data only: inputs + the reference's outputs).  No reference
source travels.  The fixtures pin ``oracle/cpu_ref.py`` (CPU tests) and the HIP path (GPU
tests).  Contents:

* ``ckpt_bottomup.npz`` / ``ckpt_single_instance.npz``: the reference's own fixture
  checkpoints (tests/assets/model_ckpts/minimal_instance_*) re-saved as named arrays, the
  uint8 input frames stored in the reference's own goldens
  (tests/inference/parity_golden/{bottomup,single_instance}.pkl), the head outputs of the
  reference ``Model`` on them, and the golden keypoints/scores those pickles hold.
* ``unet_tiny_*.npz``: random-weight tiny UNets (bilinear / transposed-conv / 13-node
  bottom-up) run through the reference ``Model`` incl. selected intermediate activations.
* ``peaks.npz``: tests/assets/inference/minimal_cms.pt + randomized maps (ties, negatives,
  borders) with the outputs of the reference find_local_peaks / find_global_peaks.
* ``paf.npz``: randomized PAF/peak sets with the outputs of the reference candidate
  enumeration, line scoring, matching and grouping.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness as rh  # noqa: E402

rh.install()
from sleap_nn.architectures.model import Model  # noqa: E402
from sleap_nn.inference.ops import paf as rpaf  # noqa: E402
from sleap_nn.inference.ops import peaks as rpeaks  # noqa: E402

REF = rh.REFERENCE_ROOT
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


def _np(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def ragged(prefix, lst):
    """Encode a ragged list of arrays as concatenation + offsets."""
    lst = [np.asarray(a) for a in lst]
    lens = np.array([a.shape[0] for a in lst], dtype=np.int64)
    cat = np.concatenate(lst, axis=0) if lst else np.zeros(0)
    return {prefix + "_cat": cat, prefix + "_len": lens}


# ---------------------------------------------------------------------------------------
def ckpt_fixture(kind: str, model_type: str, n_frames: int):
    d = f"{REF}/tests/assets/model_ckpts/minimal_instance_{kind}"
    cfg = yaml.safe_load(open(f"{d}/training_config.yaml"))
    bb = cfg["model_config"]["backbone_config"]["unet"]
    heads = cfg["model_config"]["head_configs"][model_type]
    m = Model("unet", rh.attrdict(bb), rh.attrdict(heads), model_type).eval()
    sd = rh.load_lightning_ckpt_state(f"{d}/best.ckpt")
    m.load_state_dict(sd, strict=True)
    gold = rh.load_pickle_tolerant(f"{REF}/tests/inference/parity_golden/{kind}.pkl")
    b0 = gold[0]
    img = b0["image"][:n_frames]
    with torch.inference_mode():
        x = torch.from_numpy(img).squeeze(1).float() / 255
        out = m(x)
    arrs = {"w/" + k: _np(v) for k, v in sd.items()}
    arrs["image"] = img
    arrs["eff_scale"] = b0["eff_scale"][:n_frames]
    for k, v in out.items():
        arrs["out/" + k] = _np(v)
    arrs["config_json"] = np.array(json.dumps({"backbone": bb, "heads": heads, "model_type": model_type, "preprocessing": cfg["data_config"]["preprocessing"]}))
    if kind == "bottomup":
        arrs.update(ragged("gold_peaks", [np.asarray(a).reshape(-1, 4) for a in b0["pred_instance_peaks"][:n_frames]]))
        arrs.update(ragged("gold_vals", [np.asarray(a) for a in b0["pred_peak_values"][:n_frames]]))
        arrs.update(ragged("gold_scores", [np.asarray(a) for a in b0["instance_scores"][:n_frames]]))
    else:
        arrs["gold_peaks"] = b0["pred_instance_peaks"][:n_frames]
        arrs["gold_vals"] = b0["pred_peak_values"][:n_frames]
    save(f"ckpt_{kind}.npz", **arrs)


# ---------------------------------------------------------------------------------------
def tiny_unet(name, bb, heads, model_type, hw, batch, seed, in_dtype="uint8"):
    torch.manual_seed(seed)
    m = Model("unet", rh.attrdict(bb), rh.attrdict(heads), model_type).eval()
    with torch.no_grad():
        for p_name, p in m.named_parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_uniform_(p)
            else:
                p.uniform_(-0.1, 0.1)
    g = torch.Generator().manual_seed(seed + 1)
    img = torch.randint(0, 256, (batch, 1, bb["in_channels"], hw[0], hw[1]), dtype=torch.uint8, generator=g)
    acts = {}
    hooks = []
    for mod_name, mod in m.named_modules():
        if isinstance(mod, (torch.nn.ReLU,)):
            continue
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and "head_layers" not in mod_name:
            hooks.append(mod.register_forward_hook(lambda mo, i, o, n=mod_name: acts.__setitem__(n, torch.relu(o).detach().clone())))
    with torch.inference_mode():
        out = m(img.squeeze(1).float() / 255)
    for h in hooks:
        h.remove()
    arrs = {"w/" + k: _np(v) for k, v in m.state_dict().items()}
    arrs["image"] = _np(img)
    for k, v in out.items():
        arrs["out/" + k] = _np(v)
    keep = list(acts.keys())
    pick = [keep[0], keep[1], keep[len(keep) // 2], keep[-1]]
    for k in pick:
        arrs["act/" + k] = _np(acts[k])
    arrs["config_json"] = np.array(json.dumps({"backbone": bb, "heads": heads, "model_type": model_type}))
    save(name, **arrs)


# ---------------------------------------------------------------------------------------
def peaks_fixture():
    arrs = {}
    cms = torch.load(f"{REF}/tests/assets/inference/minimal_cms.pt", map_location="cpu")
    arrs["minimal_cms"] = _np(cms)
    cases = {"minimal": cms.unsqueeze(0) if cms.dim() == 3 else cms}
    g = torch.Generator().manual_seed(7)
    # smooth random maps with several peaks, negatives and border maxima
    r = torch.randn(3, 5, 37, 53, generator=g)
    r = torch.nn.functional.avg_pool2d(r, 3, 1, 1) * 2.0
    r[0, 0, 0, 0] = 3.0  # corner peak
    r[1, 2, 36, 52] = 2.5  # opposite corner
    r[2, 4, 0, 20] = 2.0  # top edge
    cases["random"] = r
    # plateau ties: equal neighbours must NOT be peaks (strict >)
    t = torch.zeros(1, 2, 16, 16)
    t[0, 0, 4, 4] = 1.0
    t[0, 0, 4, 5] = 1.0
    t[0, 0, 10, 10] = 0.9
    t[0, 1, 8, 8] = 0.21
    t[0, 1, 8, 3] = 0.2  # == threshold -> dropped (strict >)
    cases["ties"] = t
    # quantised map: many exact ties for the global argmax (x/y independence)
    q = (torch.rand(2, 3, 20, 24, generator=g) * 4).floor() / 4
    cases["quant"] = q
    cases["allbelow"] = torch.full((1, 2, 8, 8), 0.05)
    for cname, c in cases.items():
        arrs[f"{cname}/cms"] = _np(c)
        for ref in (None, "integral"):
            tag = "none" if ref is None else "integral"
            for thr in (0.2,) if cname != "random" else (0.2, 0.5):
                p, v, s, ch = rpeaks.find_local_peaks(c, threshold=thr, refinement=ref, integral_patch_size=5)
                key = f"{cname}/local_{tag}_{thr}"
                arrs[key + "/pts"], arrs[key + "/vals"], arrs[key + "/sb"], arrs[key + "/sc"] = _np(p), _np(v), _np(s), _np(ch)
                gp, gv = rpeaks.find_global_peaks(c, threshold=thr, refinement=ref, integral_patch_size=5)
                key = f"{cname}/global_{tag}_{thr}"
                arrs[key + "/pts"], arrs[key + "/vals"] = _np(gp), _np(gv)
    # patch size 3 and 7 on the random case
    for ps in (3, 7):
        p, v, s, ch = rpeaks.find_local_peaks(cases["random"], threshold=0.2, refinement="integral", integral_patch_size=ps)
        arrs[f"random/local_integral_p{ps}/pts"] = _np(p)
    save("peaks.npz", **arrs)


# ---------------------------------------------------------------------------------------
def paf_fixture():
    sys.path.insert(0, ROOT)
    from oracle import cpu_ref  # only for the synthetic renderers (inputs, not expectations)

    arrs = {}
    specs = [
        # name, n_nodes, edges, size, n_inst, cm stride, paf stride, seed
        ("chain5", 5, [(i, i + 1) for i in range(4)], 256, 3, 2, 4, 11),
        ("tree6", 6, [(0, 1), (0, 2), (2, 3), (2, 4), (4, 5)], 320, 4, 4, 8, 12),
        ("chain13", 13, [(i, i + 1) for i in range(12)], 384, 5, 4, 8, 13),
        ("rev4", 4, [(2, 3), (1, 2), (0, 1)], 256, 3, 2, 4, 14),  # edge list not in BFS order
    ]
    meta = {}
    for name, n_nodes, edges, size, n_inst, cs, ps, seed in specs:
        bsz = 2
        cms, pafs = [], []
        for b in range(bsz):
            pts = cpu_ref.render_instances(size, n_nodes, n_inst, seed * 100 + b)
            cms.append(cpu_ref.render_confmaps(pts, size, cs, 2.5 * cs / 2))
            pafs.append(cpu_ref.render_pafs(pts, edges, size, ps, 12.0))
        cms = torch.stack(cms)
        pafs = torch.stack(pafs)
        g = torch.Generator().manual_seed(seed)
        cms = cms + 0.02 * torch.randn(cms.shape, generator=g)
        pafs = pafs + 0.05 * torch.randn(pafs.shape, generator=g)
        names = [f"n{i}" for i in range(n_nodes)]
        scorer = rpaf.PAFScorer(part_names=names, edges=[(names[s], names[d]) for s, d in edges], pafs_stride=ps, min_instance_peaks=0)
        p, v, sb, sc = rpeaks.find_local_peaks(cms, threshold=0.2, refinement="integral", integral_patch_size=5)
        p = p * cs
        pk = [p[sb == b] for b in range(bsz)]
        pv = [v[sb == b] for b in range(bsz)]
        pc = [sc[sb == b] for b in range(bsz)]
        pafs_hwc = pafs.permute(0, 2, 3, 1)
        e, ep, ls = scorer.score_paf_lines(pafs_hwc, pk, pc)
        me, ms, md, ml = scorer.match_candidates(e, ep, ls)
        inst, ivals, iscores = scorer.group_instances(pk, pv, pc, me, ms, md, ml)
        arrs[f"{name}/cms"] = _np(cms)
        arrs[f"{name}/pafs"] = _np(pafs)
        arrs.update(ragged(f"{name}/peaks", [_np(x) for x in pk]))
        arrs.update(ragged(f"{name}/vals", [_np(x) for x in pv]))
        arrs.update(ragged(f"{name}/chans", [_np(x) for x in pc]))
        arrs.update(ragged(f"{name}/edge_inds", [_np(x) for x in e]))
        arrs.update(ragged(f"{name}/edge_peak_inds", [_np(x) for x in ep]))
        arrs.update(ragged(f"{name}/line_scores", [_np(x) for x in ls]))
        arrs.update(ragged(f"{name}/match_edge", [_np(x) for x in me]))
        arrs.update(ragged(f"{name}/match_src", [_np(x) for x in ms]))
        arrs.update(ragged(f"{name}/match_dst", [_np(x) for x in md]))
        arrs.update(ragged(f"{name}/match_score", [_np(x) for x in ml]))
        arrs.update(ragged(f"{name}/inst", [_np(x).reshape(x.shape[0], -1) for x in inst]))
        arrs.update(ragged(f"{name}/inst_vals", [_np(x) for x in ivals]))
        arrs.update(ragged(f"{name}/inst_scores", [_np(x) for x in iscores]))
        arrs[f"{name}/sorted_edge_inds"] = np.array(scorer.sorted_edge_inds, dtype=np.int32)
        meta[name] = {"n_nodes": n_nodes, "edges": edges, "cms_stride": cs, "pafs_stride": ps, "size": size}
        print(name, "peaks", [x.shape[0] for x in pk], "cands", [x.shape[0] for x in e], "inst", [x.shape[0] for x in inst])
    # line_subs KAT on awkward coordinates (.5 rounding, negatives, clipping)
    peaks_s = torch.tensor([[0.0, 0.0], [5.0, 9.0], [3.0, 1.0], [-2.0, 30.0], [7.0, 7.0], [1.0, 3.0]])
    epi = torch.tensor([[0, 1], [2, 3], [4, 5], [5, 4], [1, 1]])
    ei = torch.tensor([0, 1, 0, 1, 0], dtype=torch.int32)
    ls_ = rpaf.make_line_subs(peaks_s, epi, ei, n_line_points=10, pafs_stride=2, pafs_hw=(12, 6))
    arrs["linesubs/peaks"], arrs["linesubs/epi"], arrs["linesubs/ei"], arrs["linesubs/out"] = _np(peaks_s), _np(epi), _np(ei), _np(ls_)
    # toposort cases
    topo = {}
    for nm, edges in {"chain": [(0, 1), (1, 2), (2, 3)], "rev": [(2, 3), (1, 2), (0, 1)], "tree": [(0, 1), (0, 2), (2, 3), (2, 4), (4, 5)], "star": [(3, 0), (3, 1), (3, 2)], "forest": [(0, 1), (2, 3)], "mix": [(1, 2), (0, 1), (1, 3), (3, 4)]}.items():
        et = [rpaf.EdgeType(s, d) for s, d in edges]
        topo[nm] = {"edges": edges, "order": list(rpaf.toposort_edges(et))}
    arrs["meta_json"] = np.array(json.dumps({"specs": meta, "toposort": topo}))
    save("paf.npz", **arrs)


def evaluation_fixture():
    """Pose metrics of the reference (sleap_nn/evaluation.py) on random frames: compute_oks, match_instances / compute_dists through
    duck-typed frames (the functions only touch .instances / .numpy() / .score / .frame_idx), and the Evaluator's own
    voc_metrics / mOKS / distance_metrics / pck_metrics methods on those matches."""
    import importlib

    ev = importlib.import_module("sleap_nn.evaluation")

    class Inst:
        def __init__(self, pts, score=None):
            self._p = pts
            if score is not None:
                self.score = score

        def numpy(self):
            return self._p

    class Frame:
        def __init__(self, idx, insts):
            self.frame_idx, self.instances, self.video = idx, insts, None

    rng = np.random.RandomState(12)
    arrs = {}
    n_frames, N = 12, 7
    pairs = []
    for f in range(n_frames):
        n_gt = rng.randint(1, 5)
        gt = rng.uniform(20, 300, size=(n_gt, 1, 2)) + rng.normal(0, 25, size=(n_gt, N, 2))
        gt[rng.rand(n_gt, N) < 0.1] = np.nan
        n_pr = max(1, n_gt + rng.randint(-1, 2))
        pr = np.stack([gt[i % n_gt] + rng.normal(0, [0.5, 2.0, 6.0][rng.randint(3)], size=(N, 2)) for i in range(n_pr)])
        pr[rng.rand(n_pr, N) < 0.08] = np.nan
        if f == 3:
            pr[0] = gt[0] + 400  # a far-off prediction: no match above the threshold
        scores = rng.uniform(0.2, 1.0, size=n_pr)
        gt, pr = gt.astype(np.float32), pr.astype(np.float32)
        arrs[f"f{f}/gt"], arrs[f"f{f}/pr"], arrs[f"f{f}/scores"] = gt, pr, scores
        # one prediction at a time: the reference masks invisible ground truth with a (n_gt, 1, n_nodes) boolean index, which only
        # fits n_pr == 1 -- the way match_instances calls it
        arrs[f"f{f}/oks"] = np.concatenate([ev.compute_oks(gt, pr[j : j + 1]) for j in range(n_pr)], axis=1)
        arrs[f"f{f}/oks_paper"] = np.concatenate([ev.compute_oks(gt, pr[j : j + 1], use_cocoeval=False, stddev=0.05, scale=900.0) for j in range(n_pr)], axis=1)
        pairs.append((Frame(f, [Inst(g) for g in gt]), Frame(f, [Inst(p, float(s)) for p, s in zip(pr, scores)])))
    arrs["n_frames"] = np.array(n_frames)
    E = ev.Evaluator.__new__(ev.Evaluator)
    E.positive_pairs, E.false_negatives = ev.match_frame_pairs(pairs, stddev=0.025, scale=None, threshold=0)
    E.dists_dict = ev.compute_dists(E.positive_pairs)
    arrs["pair_oks"] = np.array([o for _, _, o in E.positive_pairs])
    arrs["n_false_negatives"] = np.array(len(E.false_negatives))
    arrs["dists"] = E.dists_dict["dists"]
    voc = E.voc_metrics()
    for k in ("oks_voc.AP", "oks_voc.AR", "oks_voc.mAP", "oks_voc.mAR", "oks_voc.precisions", "oks_voc.match_scores"):
        arrs["voc/" + k] = np.asarray(voc[k])
    arrs["mOKS"] = np.array(E.mOKS()["mOKS"])
    dm = E.distance_metrics()
    arrs["dist/summary"] = np.array([dm[k] for k in ("avg", "p50", "p75", "p90", "p95", "p99")])
    pk = E.pck_metrics()
    arrs["pck/summary"] = np.array([pk["mPCK"], pk["PCK@5"], pk["PCK@10"]])
    arrs["pck/mPCK_parts"] = pk["mPCK_parts"]
    save("evaluation.npz", **arrs)


def stem_and_kernel_fixtures():
    """UNet variants outside the hot-path configs that a reference checkpoint may use: a stem block (stem_stride 2: 7x7 stem convs,
    the deepest feature at 2 x max_stride, the stem output as the last skip) and kernel_size 5."""
    bb = {"in_channels": 1, "kernel_size": 3, "filters": 8, "filters_rate": 1.5, "max_stride": 8, "stem_stride": 2, "middle_block": True, "up_interpolate": True,
          "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    heads = {"confmaps": {"part_names": ["a", "b", "c"], "sigma": 1.5, "output_stride": 2, "loss_weight": 1.0}}
    tiny_unet("unet_tiny_stem.npz", bb, heads, "single_instance", (64, 96), 2, seed=41)
    bb5 = dict(bb, kernel_size=5, stem_stride=None, in_channels=3, up_interpolate=False)
    heads5 = {"confmaps": {"part_names": ["a", "b"], "sigma": 1.5, "output_stride": 2, "loss_weight": 1.0},
              "pafs": {"edges": [["a", "b"]], "sigma": 4.0, "output_stride": 4, "loss_weight": 1.0}}
    tiny_unet("unet_tiny_k5.npz", bb5, heads5, "bottomup", (40, 56), 2, seed=43)


def winograd_kernel_fixture():
    """A filters = 16 UNet (channels 16 / 32 / 64 / 128 / 256, bottom-up heads) small enough to commit: the reference Model's outputs and a
    few activations for the layers the MI355X build runs on its Winograd F(2x2,3x3) kernels -- 16 -> 32 and 32 -> 32 + pool (wave-private
    kernel), 32 -> 64 ... 256 -> 256 and the two-source decoder convs (wave-split kernel) -- on a frame size that cuts their tiles."""
    bb = {"in_channels": 1, "kernel_size": 3, "filters": 16, "filters_rate": 2, "max_stride": 8, "stem_stride": None, "middle_block": True, "up_interpolate": True,
          "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    heads = {"confmaps": {"part_names": ["a", "b", "c", "d"], "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0},
             "pafs": {"edges": [["a", "b"], ["b", "c"], ["c", "d"]], "sigma": 15, "output_stride": 4, "loss_weight": 1.0}}
    tiny_unet("unet_f16_wino.npz", bb, heads, "bottomup", (56, 88), 2, seed=47)


def core_fixtures():
    ckpt_fixture("bottomup", "bottomup", n_frames=2)
    ckpt_fixture("single_instance", "single_instance", n_frames=2)
    tiny_unet(
        "unet_tiny_interp.npz",
        {"in_channels": 1, "kernel_size": 3, "filters": 8, "filters_rate": 2, "max_stride": 8, "stem_stride": None, "middle_block": True, "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 2},
        {"confmaps": {"part_names": ["a", "b", "c", "d", "e"], "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0}},
        "single_instance", (64, 96), 2, 101,
    )
    tiny_unet(
        "unet_tiny_trans.npz",
        {"in_channels": 1, "kernel_size": 3, "filters": 8, "filters_rate": 1.5, "max_stride": 8, "stem_stride": None, "middle_block": True, "up_interpolate": False, "stacks": 1, "convs_per_block": 2, "output_stride": 2},
        {"confmaps": {"part_names": ["a", "b", "c"], "sigma": 2.5, "output_stride": 2, "loss_weight": 1.0}, "pafs": {"edges": [["a", "b"], ["b", "c"]], "sigma": 15, "output_stride": 4, "loss_weight": 1.0}},
        "bottomup", (48, 80), 2, 102,
    )
    tiny_unet(
        "unet_tiny_bu13.npz",
        {"in_channels": 1, "kernel_size": 3, "filters": 4, "filters_rate": 2, "max_stride": 32, "stem_stride": None, "middle_block": True, "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 4},
        {"confmaps": {"part_names": [f"n{i}" for i in range(13)], "sigma": 2.5, "output_stride": 4, "loss_weight": 1.0}, "pafs": {"edges": [[f"n{i}", f"n{i+1}"] for i in range(12)], "sigma": 75, "output_stride": 8, "loss_weight": 1.0}},
        "bottomup", (64, 96), 1, 103,
    )
    tiny_unet(
        "unet_tiny_rgb.npz",
        {"in_channels": 3, "kernel_size": 3, "filters": 8, "filters_rate": 2, "max_stride": 4, "stem_stride": None, "middle_block": True, "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 1},
        {"confmaps": {"part_names": ["a", "b"], "sigma": 2.5, "output_stride": 1, "loss_weight": 1.0}},
        "single_instance", (32, 40), 1, 104,
    )
    peaks_fixture()
    paf_fixture()


# ---------------------------------------------------------------------------------------
def _load_ckpt_model(kind, model_type):
    d = f"{REF}/tests/assets/model_ckpts/minimal_instance_{kind}"
    cfg = yaml.safe_load(open(f"{d}/training_config.yaml"))
    bb = cfg["model_config"]["backbone_config"]["unet"]
    heads = cfg["model_config"]["head_configs"][model_type]
    m = Model("unet", rh.attrdict(bb), rh.attrdict(heads), model_type).eval()
    sd = rh.load_lightning_ckpt_state(f"{d}/best.ckpt")
    m.load_state_dict(sd, strict=True)
    return m, sd, bb, heads, cfg


def topdown_fixture():
    """Centroid + centered-instance fixture checkpoints on the frames of the reference's own
    tests/inference/parity_golden/topdown.pkl, stage by stage through the reference ops."""
    from sleap_nn.inference.ops.crops import crop_bboxes, make_centered_bboxes

    mc, sdc, bbc, hc, cfgc = _load_ckpt_model("centroid", "centroid")
    mi, sdi, bbi, hi, cfgi = _load_ckpt_model("centered_instance", "centered_instance")
    gold = rh.load_pickle_tolerant(f"{REF}/tests/inference/parity_golden/topdown.pkl")
    frames = np.stack([gold[0]["image"][0], gold[1]["image"][0]])  # two distinct frames (rows of a batch repeat the frame)
    arrs = {"wc/" + k: _np(v) for k, v in sdc.items()}
    arrs.update({"wi/" + k: _np(v) for k, v in sdi.items()})
    arrs["image"] = frames
    with torch.inference_mode():
        img = torch.from_numpy(frames)
        cms = mc(img.float() / 255)["CentroidConfmapsHead"]
        # thresholds of the reference's golden capture (tests/utils/parity_goldens.py:125-134): 0.03, max_instances 6
        p, v, sb, _ = rpeaks.find_local_peaks(cms, threshold=0.03, refinement="integral", integral_patch_size=5)
        p = p * hc["confmaps"]["output_stride"]
        B = frames.shape[0]
        mi_ = 6
        cent = torch.full((B, mi_, 2), float("nan"))
        cval = torch.full((B, mi_), float("nan"))
        for b in range(B):
            m = sb == b
            pb, vb = p[m], v[m]
            if pb.shape[0] > mi_:
                vb, ix = torch.topk(vb, mi_)
                pb = pb[ix]
            cent[b, : pb.shape[0]] = pb
            cval[b, : pb.shape[0]] = vb
        valid = ~torch.isnan(cent).any(-1)
        idx = valid.nonzero()
        vc = cent[idx[:, 0], idx[:, 1]]
        crop = int(cfgi["data_config"]["preprocessing"]["crop_size"])
        bboxes = make_centered_bboxes(vc, crop, crop)
        crops = crop_bboxes(img, bboxes, idx[:, 0])
        cms2 = mi(crops.float() / 255)["CenteredInstanceConfmapsHead"]
        gp, gv = rpeaks.find_global_peaks(cms2, threshold=0.03, refinement="integral", integral_patch_size=5)
        gp = gp * hi["confmaps"]["output_stride"]
    # cross-check against the reference's own golden (first frame, its instances)
    g0 = gold[0]
    d = np.abs(np.nan_to_num(_np(cent[0]), nan=1e9)[:, None, :] - g0["pred_centroids"][None, :, :]).sum(-1)
    assert d.min(axis=0).max() < 1e-3, d
    arrs.update({"out/CentroidConfmapsHead": _np(cms), "centroids": _np(cent), "centroid_vals": _np(cval), "valid_idx": _np(idx), "bboxes": _np(bboxes),
                 "crops": _np(crops), "out/CenteredInstanceConfmapsHead": _np(cms2), "crop_peaks": _np(gp), "crop_peak_vals": _np(gv),
                 "gold0_pred_centroids": g0["pred_centroids"], "gold0_pred_instance_peaks": g0["pred_instance_peaks"], "gold0_instance_image": g0["instance_image"],
                 "gold0_instance_bbox": g0["instance_bbox"]})
    arrs["config_json"] = np.array(json.dumps({"centroid": {"backbone": bbc, "heads": hc, "model_type": "centroid"}, "centered": {"backbone": bbi, "heads": hi, "model_type": "centered_instance"}, "crop_size": crop}))
    # randomized crop KAT (fractional, negative and out-of-frame corners; float and uint8 images)
    g = torch.Generator().manual_seed(3)
    im = torch.randint(0, 256, (3, 2, 40, 52), dtype=torch.uint8, generator=g)
    pts = torch.tensor([[5.3, 4.1], [0.0, 0.0], [51.0, 39.0], [-3.6, 12.2], [25.5, -2.5], [60.0, 20.0], [26.49, 20.51]])
    si = torch.tensor([0, 1, 2, 0, 1, 2, 1])
    for hw in ((8, 8), (7, 11), (16, 12)):
        bb = make_centered_bboxes(pts, hw[0], hw[1])
        arrs[f"cropkat/{hw[0]}x{hw[1]}/u8"] = _np(crop_bboxes(im, bb, si))
        arrs[f"cropkat/{hw[0]}x{hw[1]}/f32"] = _np(crop_bboxes(im.float() / 7, bb, si))
        arrs[f"cropkat/{hw[0]}x{hw[1]}/bboxes"] = _np(bb)
    arrs["cropkat/image"], arrs["cropkat/pts"], arrs["cropkat/si"] = _np(im), _np(pts), _np(si)
    save("topdown.npz", **arrs)


def topdown_sized_fixture():
    """The reference's own TopDownLayer.predict (layers/topdown.py:36-466) on the fixture checkpoints WITH the centroid layer's sizematcher active (frames resized and
    padded to max_height x max_width: eff_scale != 1): pins the sized-space handling of stage 2 -- boxes around centroid * eff_scale, crops cut from the sizematched
    frame, keypoints and boxes divided by eff_scale (topdown.py:127-150, 262-267) -- which the stage-by-stage `topdown.npz` (eff_scale = 1) cannot see."""
    import torch.nn as nn

    from sleap_nn.inference.layers.backends.torch_backend import TorchBackend
    from sleap_nn.inference.layers.centered_instance import CenteredInstanceLayer
    from sleap_nn.inference.layers.centroid import CentroidLayer
    from sleap_nn.inference.layers.configs import PostprocessConfig, PreprocessConfig
    from sleap_nn.inference.layers.topdown import TopDownLayer

    class Fwd(nn.Module):  # the LightningModule forward preamble (lightning_modules.py:1840-1848): squeeze the n_samples axis, normalize
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, x):
            x = torch.squeeze(x, dim=1)
            if x.dtype == torch.uint8 or x.max() > 1.0:
                x = x.float() / 255.0
            return self.m(x.float())

    # torchvision (pyproject.toml:40, torchvision>=0.20.0) is absent from this image and stubbed by the harness; the ONE call the sizematcher makes into it,
    # transforms.v2.functional.resize on a tensor, is the torch operator interpolate(bilinear, antialias=True) (uint8 natively) -- the statement oracle/cpu_ref.py:1221 and
    # tests/test_oracle_golden.py:323 already rest on.  Everything else below is the reference's own code.
    import torch.nn.functional as F

    import sleap_nn.data.resizing as rresizing

    def tv_resize(image, size, **_kw):
        x = image if image.dim() == 4 else image[None]
        y = F.interpolate(x if x.dtype == torch.uint8 else x.float(), size=tuple(size), mode="bilinear", align_corners=False, antialias=True)
        return y if image.dim() == 4 else y[0]

    rresizing.tvf.resize = tv_resize
    mc, sdc, bbc, hc, cfgc = _load_ckpt_model("centroid", "centroid")
    mi, sdi, bbi, hi, cfgi = _load_ckpt_model("centered_instance", "centered_instance")
    gold = rh.load_pickle_tolerant(f"{REF}/tests/inference/parity_golden/topdown.pkl")
    frames = np.stack([gold[0]["image"][0], gold[1]["image"][0]])
    H, W = frames.shape[-2:]
    crop = int(cfgi["data_config"]["preprocessing"]["crop_size"])
    arrs = {"image": frames}
    # (the fixture model is not scale-invariant: a mild up-scaling and a mild down-scaling, both height-bound fits with padding on the right)
    # The reference infers the crop size from its FIRST box in float32 (ops/crops.py:66-67: int(|y_bl - y_tl|) + 1), which comes out one short of crop_size for about half
    # of all fractional box centres; with return_crops=True such a batch raises in its scatter (topdown.py:300-310).  The candidates are tried in turn and the first up-
    # and the first down-scaling whose crops have the configured size are kept -- the product always cuts crop_size x crop_size.
    candidates = {"up": [(H + 64, W + 96), (H + 48, W + 64), (H + 96, W + 128), (H + 32, W + 64), (H + 80, W + 96)], "down": [(H - 32, W), (H - 48, W), (H - 16, W), (H - 64, W - 32)]}
    cases = {}
    for tag, (mh, mw) in [(t, c) for t, cs in candidates.items() for c in cs]:
        if tag in cases:
            continue
        cl = CentroidLayer(TorchBackend(Fwd(mc), device="cpu"), hc["confmaps"]["output_stride"], max_instances=6, max_stride=bbc["max_stride"],
                           preprocess_config=PreprocessConfig(max_height=mh, max_width=mw), postprocess_config=PostprocessConfig(peak_threshold=0.03, max_instances=6))
        il = CenteredInstanceLayer(TorchBackend(Fwd(mi), device="cpu"), hi["confmaps"]["output_stride"], max_stride=bbi["max_stride"],
                                   postprocess_config=PostprocessConfig(peak_threshold=0.03))
        td = TopDownLayer(cl, il, (crop, crop), return_crops=True)
        try:
            with torch.inference_mode():
                out = td.predict(torch.from_numpy(frames))
        except RuntimeError as e:
            print(f"topdown_sized[{tag}] max {mh} x {mw}: reference raised ({str(e)[:80]}...) -- next candidate")
            continue
        if int((~torch.isnan(out.pred_centroids[..., 0])).sum()) < 2:
            continue
        cases[tag] = (mh, mw)
        eff = out.preprocess_info.eff_scale if getattr(out, "preprocess_info", None) is not None else None
        arrs[f"{tag}/max_hw"] = np.array([mh, mw])
        for f in ("pred_keypoints", "pred_crop_keypoints", "pred_peak_values", "pred_centroids", "pred_centroid_values", "instance_bboxes", "crops", "instance_scores"):
            v = getattr(out, f, None)
            if v is not None:
                arrs[f"{tag}/{f}"] = _np(v)
        n = int((~torch.isnan(out.pred_centroids[..., 0])).sum())
        print(f"topdown_sized[{tag}]: max {mh} x {mw}, {n} instances, eff_scale {None if eff is None else _np(eff)}")
    assert set(cases) == {"up", "down"}, cases
    save("topdown_sized.npz", **arrs)


def multiclass_fixture():
    from sleap_nn.inference.ops.identity import classify_peaks_from_maps

    m, sd, bb, heads, cfg = _load_ckpt_model("multiclass_bottomup", "multi_class_bottomup")
    gold = rh.load_pickle_tolerant(f"{REF}/tests/inference/parity_golden/multiclass_bottomup.pkl")
    b0 = gold[0]
    img = b0["image"][:3]
    arrs = {"w/" + k: _np(v) for k, v in sd.items()}
    arrs["image"] = img
    with torch.inference_mode():
        out = m(torch.from_numpy(img).squeeze(1).float() / 255)
        cms, cmaps = out["MultiInstanceConfmapsHead"], out["ClassMapsHead"]
        cs, ks = heads["confmaps"]["output_stride"], heads["class_maps"]["output_stride"]
        p, v, sb, sc = rpeaks.find_local_peaks(cms, threshold=0.05, refinement="integral", integral_patch_size=5)  # parity_goldens.py:141-146
        p = p * cs
        inst, pv, cp = classify_peaks_from_maps(cmaps, p / ks, v, sb, sc, n_channels=cms.shape[1])
        inst = inst * ks / cfg["data_config"]["preprocessing"]["scale"]
    gk = np.stack([np.asarray(a) for a in b0["pred_instance_peaks"][:3]])
    assert np.allclose(_np(inst), gk, atol=1e-3, equal_nan=True), (inst, gk)  # reproduces the reference golden
    for k, t in out.items():
        arrs["out/" + k] = _np(t)
    arrs.update({"inst": _np(inst), "peak_vals": _np(pv), "class_probs": _np(cp), "gold_peaks": gk, "gold_vals": b0["pred_peak_values"][:3]})
    arrs["config_json"] = np.array(json.dumps({"backbone": bb, "heads": heads, "model_type": "multi_class_bottomup", "preprocessing": cfg["data_config"]["preprocessing"]}))
    # randomized KAT with many peaks per (sample, node) and probability ties
    g = torch.Generator().manual_seed(11)
    cm = torch.rand((2, 3, 24, 30), generator=g)
    cm = (cm * 8).round() / 8  # ties
    n = 60
    pts = torch.rand((n, 2), generator=g) * torch.tensor([34.0, 28.0]) - 2.0
    pts[::7] = pts[::7].round() + 0.5  # exact .5 -> round-half-even
    vals = torch.rand((n,), generator=g)
    sb = torch.randint(0, 2, (n,), generator=g).to(torch.int32).sort().values
    sc = torch.randint(0, 4, (n,), generator=g).to(torch.int32)
    ki, kv, kp = classify_peaks_from_maps(cm, pts, vals, sb, sc, n_channels=4)
    arrs.update({"kat/class_maps": _np(cm), "kat/pts": _np(pts), "kat/vals": _np(vals), "kat/sb": _np(sb), "kat/sc": _np(sc), "kat/points": _np(ki), "kat/point_vals": _np(kv), "kat/class_probs": _np(kp)})
    save("multiclass.npz", **arrs)





def targets_fixture():
    """Reference target rendering (generate_multiconfmaps / generate_pafs) on random instance sets."""
    from sleap_nn.data.confidence_maps import generate_multiconfmaps
    from sleap_nn.data.edge_maps import generate_pafs

    arrs = {}
    g = torch.Generator().manual_seed(17)
    hw = (96, 128)
    edges = [(0, 1), (1, 2), (1, 3)]
    pts = torch.rand((3, 4, 4, 2), generator=g) * torch.tensor([140.0, 110.0]) - 8.0  # some out of frame
    pts[0, 1, 2] = float("nan")  # missing node
    pts[1, 3] = float("nan")  # missing instance
    pts[2, 0] = torch.tensor([[-5.0, -5.0], [-3.0, 200.0], [300.0, 2.0], [-1.0, -1.0]])  # instance fully outside
    pts[2, 1, 1] = pts[2, 1, 0]  # zero-length edge
    arrs["points"] = _np(pts)
    for stride, sigma in ((2, 1.5), (4, 2.5)):
        cm = torch.cat([generate_multiconfmaps(pts[b : b + 1], hw, num_instances=4, sigma=sigma, output_stride=stride) for b in range(3)])
        arrs[f"confmaps_s{stride}"] = _np(cm)
    for stride, sigma in ((4, 15.0), (8, 50.0)):
        pf = torch.stack([generate_pafs(pts[b : b + 1].clone(), hw, sigma=sigma, output_stride=stride, edge_inds=torch.tensor(edges), flatten_channels=True) for b in range(3)])
        arrs[f"pafs_s{stride}"] = _np(pf)
    arrs["meta_json"] = np.array(json.dumps({"hw": hw, "edges": edges, "confmaps": [[2, 1.5], [4, 2.5]], "pafs": [[4, 15.0], [8, 50.0]]}))
    save("targets.npz", **arrs)


def convnext_decoder_fixture():
    """The importable half of ConvNextWrapper: the reference's own MaxPool2dWithSamePadding ->
    middle SimpleConvBlocks -> Decoder(encoder_channels=...) constructed exactly as
    convnext.py:216-301 does, on random encoder features.  (The encoder half needs torchvision,
    which this image does not have -- see oracle/cpu_ref.py:convnext_plan.)"""
    from sleap_nn.architectures.common import MaxPool2dWithSamePadding
    from sleap_nn.architectures.encoder_decoder import Decoder, SimpleConvBlock

    arrs = {}
    cases = {
        "a": dict(channels=[8, 16, 32, 64], stem_stride=2, output_stride=2, rate=2, cpb=2, hw=(64, 96)),
        "b": dict(channels=[8, 16, 32, 64], stem_stride=4, output_stride=1, rate=2, cpb=2, hw=(64, 64)),
        "c": dict(channels=[6, 12, 24, 48], stem_stride=2, output_stride=4, rate=1.5, cpb=3, hw=(64, 64)),
    }
    for tag, c in cases.items():
        torch.manual_seed(31 + ord(tag))
        ch, ss, os_ = c["channels"], c["stem_stride"], c["output_stride"]
        last = ch[-1]
        max_stride = ss * 8 * 2
        up_blocks = int(np.log2(max_stride / (ss * os_)).astype(int) + np.log2(ss).astype(int))
        mods = torch.nn.ModuleDict()
        middle = torch.nn.ModuleList()
        if c["cpb"] > 1:
            middle.append(SimpleConvBlock(in_channels=last, pool=False, pool_before_convs=False, pooling_stride=2, num_convs=c["cpb"] - 1,
                                          filters=int(last * c["rate"]), kernel_size=3, use_bias=True, batch_norm=False, activation="relu",
                                          prefix="convnext_middle_expand"))
        middle.append(SimpleConvBlock(in_channels=int(last * c["rate"]), pool=False, pool_before_convs=False, pooling_stride=2, num_convs=1,
                                      filters=int(last * c["rate"]), kernel_size=3, use_bias=True, batch_norm=False, activation="relu",
                                      prefix="convnext_middle_contract"))
        # convnext.py:288-301: convs_per_block is NOT forwarded, the Decoder keeps its default of 2
        dec = Decoder(x_in_shape=int(last * c["rate"]), current_stride=max_stride, filters=ch[0], up_blocks=up_blocks, down_blocks=len(ch) - 1,
                      filters_rate=c["rate"], kernel_size=3, stem_blocks=1, block_contraction=False, output_stride=os_, up_interpolate=True,
                      encoder_channels=ch[::-1])
        mods["middle_blocks"] = middle
        mods["dec"] = dec
        pool = MaxPool2dWithSamePadding(kernel_size=2, stride=2, padding="same")
        H, W = c["hw"]
        feats = [torch.randn(2, ch[i], H // (ss * 2**i), W // (ss * 2**i)) for i in range(4)]  # enc_output[::2]
        x_last = torch.randn(2, last, H // (ss * 8), W // (ss * 8))  # enc_output[-1]
        with torch.no_grad():
            x = pool(x_last)
            for blk in middle:
                x = blk(x)
            out = dec(x, feats[::-1])
        for k, v in mods.state_dict().items():
            arrs[f"{tag}/w/backbone.{k}"] = _np(v)
        for i, f in enumerate(feats):
            arrs[f"{tag}/feat{i}"] = _np(f)
        arrs[f"{tag}/x_last"] = _np(x_last)
        arrs[f"{tag}/middle"] = _np(x)
        for i, o in enumerate(out["outputs"]):
            arrs[f"{tag}/out{i}"] = _np(o)
        arrs[f"{tag}/meta_json"] = np.array(json.dumps({**c, "strides": [int(v) for v in out["strides"]],
                                                        "stride_to_filters": {int(k): int(v) for k, v in dec.stride_to_filters.items()}}))
    save("convnext_decoder.npz", **arrs)


def filters_fixture():
    """Reference NMS / IoU / OKS helpers (inference/ops/filters.py, pure numpy) on random instance sets."""
    import types

    sys.modules.setdefault("sleap_io", types.ModuleType("sleap_io"))
    from sleap_nn.inference.ops import filters as rf

    rng = np.random.default_rng(41)
    arrs = {}
    n_cases = 24
    for c in range(n_cases):
        n_inst, n_nodes = int(rng.integers(1, 9)), int(rng.integers(2, 8))
        centres = rng.uniform(20, 200, size=(n_inst, 1, 2))
        if c % 3 == 0:  # clusters of near-duplicates
            centres = centres[rng.integers(0, max(1, n_inst // 2), size=n_inst)] + rng.normal(0, 2.0, size=(n_inst, 1, 2))
        pts = centres + rng.normal(0, 15.0, size=(n_inst, n_nodes, 2))
        pts[rng.random((n_inst, n_nodes)) < 0.2] = np.nan
        if c % 5 == 0:
            pts[0] = np.nan if n_inst > 1 else pts[0]
        scores = rng.uniform(0.1, 1.0, size=n_inst)
        if c % 4 == 0 and n_inst > 1:
            scores[1] = scores[0]  # tie
        arrs[f"{c}/points"] = pts
        arrs[f"{c}/scores"] = scores
        boxes = np.array([rf._instance_bbox(types.SimpleNamespace(numpy=lambda p=p: p)) for p in pts])
        arrs[f"{c}/bboxes"] = boxes
        for thr in (0.1, 0.5, 0.8):
            arrs[f"{c}/keep_iou_{thr}"] = np.array(rf._nms_greedy_iou(boxes, scores, thr), dtype=np.int64)
            arrs[f"{c}/keep_oks_{thr}"] = np.array(rf._nms_greedy_oks([p for p in pts], scores, thr), dtype=np.int64)
        arrs[f"{c}/oks_matrix"] = np.array([[rf._compute_oks(a, b) for b in pts] for a in pts])
        arrs[f"{c}/iou_row0"] = rf._compute_iou_one_to_many(boxes[0], boxes)
    arrs["n_cases"] = np.array(n_cases)
    save("filters.npz", **arrs)


def multiclass_topdown_fixture():
    """Multi-class centered-instance fixture checkpoint (confmaps + ClassVectorsHead) on the crops stored in the
    reference's own tests/inference/parity_golden/multiclass_topdown.pkl, plus that golden's expectations."""
    from sleap_nn.inference.ops.identity import get_class_inds_from_vectors

    m, sd, bb, heads, cfg = _load_ckpt_model("multiclass_centered_instance", "multi_class_topdown")
    gold = rh.load_pickle_tolerant(f"{REF}/tests/inference/parity_golden/multiclass_topdown.pkl")
    arrs = {"w/" + k: _np(v) for k, v in sd.items()}
    frames, crops, kp, kv, cls, cent, fidx = [], [], [], [], [], [], []
    usable = [b for b in gold if b["instance_image"].shape[-2:] == (128, 128)][:3]  # some goldens hold border crops of 127 px
    for bi, b in enumerate(usable):
        frames.append(b["image"][0])
        crops.append(b["instance_image"][:, 0])
        kp.append(b["pred_instance_peaks"])
        kv.append(b["pred_peak_values"])
        cls.append(b["pred_class_inds"])
        cent.append(b["pred_centroids"])
        fidx.append(np.full(len(b["pred_class_inds"]), bi))
    crops_t = torch.from_numpy(np.concatenate(crops))
    with torch.inference_mode():
        out = m(crops_t.float() / 255)
        gp, gv = rpeaks.find_global_peaks(out["CenteredInstanceConfmapsHead"], threshold=0.03, refinement="integral", integral_patch_size=5)
        ci, cp = [], []
        fi = np.concatenate(fidx)
        for f in np.unique(fi):
            a, b_ = get_class_inds_from_vectors(out["ClassVectorsHead"][torch.from_numpy(fi == f)])
            ci.append(_np(a))
            cp.append(_np(b_))
    assert np.array_equal(np.concatenate(ci), np.concatenate(cls)), (ci, cls)
    arrs.update({"frames": np.stack(frames), "crops": _np(crops_t), "crop_frame": fi, "out/CenteredInstanceConfmapsHead": _np(out["CenteredInstanceConfmapsHead"]),
                 "out/ClassVectorsHead": _np(out["ClassVectorsHead"]), "crop_peaks": _np(gp * heads["confmaps"]["output_stride"]), "crop_peak_vals": _np(gv),
                 "class_inds": np.concatenate(ci), "class_probs": np.concatenate(cp), "gold_pred_instance_peaks": np.concatenate(kp), "gold_pred_peak_values": np.concatenate(kv),
                 "gold_pred_class_inds": np.concatenate(cls), "gold_pred_centroids": np.concatenate(cent)})
    arrs["config_json"] = np.array(json.dumps({"backbone": bb, "heads": heads, "model_type": "multi_class_topdown", "crop_size": int(cfg["data_config"]["preprocessing"]["crop_size"])}))
    save("multiclass_topdown.npz", **arrs)


def centroid_nms_fixture():
    """Reference TopDownLayer._centroid_nms_mask (layers/topdown.py:395-438) on random centroid sets."""
    import types

    from sleap_nn.inference.layers.topdown import TopDownLayer

    g = torch.Generator().manual_seed(77)
    arrs = {}
    n = 12
    for c in range(n):
        B, I = 3, int(torch.randint(1, 9, (1,), generator=g))
        crop = (int(torch.randint(16, 80, (1,), generator=g)), int(torch.randint(16, 80, (1,), generator=g)))
        thr = float([0.2, 0.5, 0.75][c % 3])
        cent = torch.rand((B, I, 2), generator=g) * 120
        if I > 2:
            cent[:, 1] = cent[:, 0] + torch.rand((B, 2), generator=g) * 12  # close pairs
        vals = torch.rand((B, I), generator=g)
        if I > 3 and c % 2 == 0:
            vals[:, 3] = vals[:, 2]  # confidence ties
        drop = torch.rand((B, I), generator=g) < 0.2
        cent[drop] = float("nan")
        vals[drop] = float("nan")
        valid = ~torch.isnan(cent).any(-1)
        ns = types.SimpleNamespace(crop_size=crop, centroid_nms_threshold=thr, _bbox_iou=TopDownLayer._bbox_iou)
        keep = TopDownLayer._centroid_nms_mask(ns, cent, vals, valid)
        arrs[f"{c}/centroids"], arrs[f"{c}/vals"], arrs[f"{c}/keep"] = _np(cent), _np(vals), _np(keep)
        arrs[f"{c}/crop"], arrs[f"{c}/thr"] = np.array(crop), np.array(thr)
    arrs["n_cases"] = np.array(n)
    save("centroid_nms.npz", **arrs)


def schedulers_fixture():
    """The reference's two closed-form schedulers (training/schedulers.py) and torch's StepLR / ReduceLROnPlateau as
    configure_optimizers builds them (lightning_modules.py:800-857): learning rate after each of 30 epochs."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_schedulers", os.path.join(rh.REFERENCE_ROOT, "sleap_nn", "training", "schedulers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    arrs = {}
    g = torch.Generator().manual_seed(5)
    losses = (torch.rand(30, generator=g) * 0.1 + torch.linspace(1.0, 0.6, 30).clamp(min=0.75)).tolist()  # plateaus after ~19 epochs

    def run(make, use_loss=False):
        p = [torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.Adam(p, lr=1e-3)
        sch = make(opt)
        out = [opt.param_groups[0]["lr"]]
        for e in range(30):
            opt.step()
            sch.step(losses[e]) if use_loss else sch.step()
            out.append(opt.param_groups[0]["lr"])
        return np.array(out, dtype=np.float64)

    arrs["cosine"] = run(lambda o: mod.LinearWarmupCosineAnnealingLR(o, warmup_epochs=4, max_epochs=25, warmup_start_lr=1e-5, eta_min=1e-6))
    arrs["cosine_nowarm"] = run(lambda o: mod.LinearWarmupCosineAnnealingLR(o, warmup_epochs=0, max_epochs=20))
    arrs["linear"] = run(lambda o: mod.LinearWarmupLinearDecayLR(o, warmup_epochs=5, max_epochs=28, warmup_start_lr=0.0, end_lr=2e-5))
    arrs["step"] = run(lambda o: torch.optim.lr_scheduler.StepLR(o, step_size=7, gamma=0.3))
    arrs["plateau_abs"] = run(lambda o: torch.optim.lr_scheduler.ReduceLROnPlateau(o, mode="min", threshold=1e-6, threshold_mode="abs", cooldown=3, patience=2, factor=0.5, min_lr=1e-8), True)
    arrs["plateau_rel"] = run(lambda o: torch.optim.lr_scheduler.ReduceLROnPlateau(o, mode="min", threshold=0.05, threshold_mode="rel", cooldown=0, patience=1, factor=0.1, min_lr=2e-5), True)
    arrs["losses"] = np.array(losses, dtype=np.float64)
    save("schedulers.npz", **arrs)


def _reference_method(path: str, cls: str, name: str, extra_globals: dict):
    """Compile ONE method of a reference class straight from its source file (the module itself cannot be imported here:
    lightning_modules.py pulls in lightning / matplotlib / wandb).  The reference's own lines run unmodified, as a function
    whose ``self`` the caller supplies; nothing is copied into this repository."""
    import ast

    src = open(path).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == name:
                    mod = ast.Module(body=[fn], type_ignores=[])
                    ns = dict(extra_globals)
                    exec(compile(mod, path, "exec"), ns)
                    return ns[name]
    raise KeyError(f"{cls}.{name} not found in {path}")


def losses_fixture():
    """the reference's loss arithmetic and one training step's gradients.
    * compute_ohkm_loss (training/losses.py:8-63) on random maps for several parameter sets (the reference has no numeric
      test of it);
    * LightningModel._compute_negative_weighted_loss (training/lightning_modules.py:490-545) for train / val stages;
    * one bottom-up training step of the reference ``Model`` (training_step :1850-1895: negative-weighted MSE + OHKM per
      head, loss-weight sum) with autograd gradients of every parameter."""
    from types import SimpleNamespace
    from typing import Dict

    from sleap_nn.training.losses import compute_ohkm_loss

    neg_loss = _reference_method(os.path.join(REF, "sleap_nn", "training", "lightning_modules.py"), "LightningModel", "_compute_negative_weighted_loss",
                                 {"torch": torch, "nn": torch.nn, "Dict": Dict})
    arrs = {}
    g = torch.Generator().manual_seed(31)
    ohkm_cases = [
        dict(hard_to_easy_ratio=2.0, min_hard_keypoints=2, max_hard_keypoints=None, loss_scale=5.0),
        dict(hard_to_easy_ratio=1.2, min_hard_keypoints=1, max_hard_keypoints=3, loss_scale=2.0),
        dict(hard_to_easy_ratio=50.0, min_hard_keypoints=4, max_hard_keypoints=None, loss_scale=1.0),
        dict(hard_to_easy_ratio=1.0, min_hard_keypoints=0, max_hard_keypoints=2, loss_scale=7.5),
    ]
    for i, kw in enumerate(ohkm_cases):
        y = torch.rand(3, 6, 14, 11, generator=g)
        p = y + torch.randn(3, 6, 14, 11, generator=g) * torch.linspace(0.02, 0.3, 6).view(1, 6, 1, 1)
        arrs[f"ohkm{i}/y"], arrs[f"ohkm{i}/p"] = _np(y), _np(p)
        arrs[f"ohkm{i}/params"] = np.array([kw["hard_to_easy_ratio"], kw["min_hard_keypoints"], -1 if kw["max_hard_keypoints"] is None else kw["max_hard_keypoints"], kw["loss_scale"]], dtype=np.float64)
        arrs[f"ohkm{i}/loss"] = np.array(float(compute_ohkm_loss(y_gt=y, y_pr=p, **kw)), dtype=np.float64)
    arrs["n_ohkm"] = np.array(len(ohkm_cases))
    y = torch.rand(5, 4, 9, 13, generator=g)
    p = y + 0.1 * torch.randn(5, 4, 9, 13, generator=g)
    neg = torch.tensor([False, True, False, True, True])
    arrs["neg/y"], arrs["neg/p"], arrs["neg/is_negative"] = _np(y), _np(p), _np(neg)
    for tag, w, stage, batch in (("w0.25_train", 0.25, "train", {"is_negative": neg}), ("w3_train", 3.0, "train", {"is_negative": neg}), ("w0.25_val", 0.25, "val", {"is_negative": neg}),
                                 ("w1_train", 1.0, "train", {"is_negative": neg}), ("absent", 0.25, "train", {})):
        arrs[f"neg/loss_{tag}"] = np.array(float(neg_loss(SimpleNamespace(negative_loss_weight=w), p, y, batch, stage=stage)), dtype=np.float64)

    # ---- one training step of the reference Model (bottom-up, bilinear decoder): loss + every parameter gradient
    bb = {"in_channels": 1, "kernel_size": 3, "filters": 8, "filters_rate": 2, "max_stride": 8, "stem_stride": None, "middle_block": True,
          "up_interpolate": True, "stacks": 1, "convs_per_block": 2, "output_stride": 2}
    names = ["a", "b", "c"]
    heads = {"confmaps": {"part_names": names, "sigma": 1.5, "output_stride": 2, "loss_weight": 1.0},
             "pafs": {"edges": [["a", "b"], ["b", "c"]], "sigma": 4.0, "output_stride": 4, "loss_weight": 0.6}}
    torch.manual_seed(77)
    m = Model("unet", rh.attrdict(bb), rh.attrdict(heads), "bottomup").train()
    with torch.no_grad():
        for _, prm in m.named_parameters():
            torch.nn.init.xavier_uniform_(prm) if prm.dim() > 1 else prm.uniform_(-0.1, 0.1)
    img = torch.randint(0, 256, (4, 1, 48, 64), dtype=torch.uint8, generator=g)
    is_neg = torch.tensor([False, True, False, False])
    ohkm = dict(hard_to_easy_ratio=1.5, min_hard_keypoints=1, max_hard_keypoints=None, loss_scale=3.0)
    nlw, lws = 0.3, [1.0, 0.6]
    self_ = SimpleNamespace(negative_loss_weight=nlw)
    preds = m(img.float() / 255.0)  # normalize_on_gpu of uint8 frames
    tg = {k: torch.rand(v.shape, generator=g) * 0.5 for k, v in preds.items()}
    hl = []
    for k in ("MultiInstanceConfmapsHead", "PartAffinityFieldsHead"):  # lightning_modules.py:1860-1895
        l = neg_loss(self_, preds[k], tg[k], {"is_negative": is_neg})
        l = l + compute_ohkm_loss(y_gt=tg[k], y_pr=preds[k], **ohkm)
        hl.append(l)
    total = sum(s_ * l for s_, l in zip(lws, hl))
    total.backward()
    for k, v in m.state_dict().items():
        arrs["step/w/" + k] = _np(v)
    for k, prm in m.named_parameters():
        arrs["step/g/" + k] = _np(prm.grad)
    arrs["step/image"] = _np(img)
    arrs["step/is_negative"] = _np(is_neg)
    for k, v in tg.items():
        arrs["step/target/" + k] = _np(v)
    arrs["step/losses"] = np.array([float(total)] + [float(l) for l in hl], dtype=np.float64)
    arrs["step/config_json"] = np.array(json.dumps({"backbone": bb, "heads": heads, "model_type": "bottomup", "negative_loss_weight": nlw, "loss_weights": lws,
                                                    "ohkm": {**ohkm, "max_hard_keypoints": -1}}))
    save("losses.npz", **arrs)


if __name__ == "__main__":
    only = sys.argv[1:]
    if not only or "losses" in only:
        losses_fixture()
    if not only or "stem" in only:
        stem_and_kernel_fixtures()
    if not only or "wino" in only:
        winograd_kernel_fixture()
    if not only or "evaluation" in only:
        evaluation_fixture()
    if not only or "schedulers" in only:
        schedulers_fixture()
    if not only or "core" in only:
        core_fixtures()
    if not only or "topdown" in only:
        topdown_fixture()
    if not only or "topdown_sized" in only:
        topdown_sized_fixture()
    if not only or "multiclass" in only:
        multiclass_fixture()
    if not only or "targets" in only:
        targets_fixture()
    if not only or "convnext" in only:
        convnext_decoder_fixture()
    if not only or "filters" in only:
        filters_fixture()
    if not only or "mctopdown" in only:
        multiclass_topdown_fixture()
    if not only or "nms" in only:
        centroid_nms_fixture()
