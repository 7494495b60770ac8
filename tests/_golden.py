"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def ragged(z, prefix):
    cat, lens = z[prefix + "_cat"], z[prefix + "_len"]
    out, o = [], 0
    for n in lens:
        out.append(cat[o : o + n])
        o += n
    return out


def config(z):
    return json.loads(str(z["config_json"]))


def weights(z):
    import torch

    return {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}
