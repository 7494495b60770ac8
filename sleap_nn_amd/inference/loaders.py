"""Checkpoint + config ingestion for existing sleap-nn run directories
(``best.ckpt`` + ``training_config.yaml``), without Lightning / OmegaConf installed.

Counterpart of ``sleap_nn/inference/loaders.py:87-221,1054`` (``load_model_assets``) reduced to
what the hot path needs: the ``state_dict`` (LightningModule keys ``model.*``), the backbone and
head configs, the preprocessing block and the skeleton's node/edge names.
"""
from __future__ import annotations

import os
import pickle
import types
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import yaml

MODEL_TYPES = ("single_instance", "centroid", "centered_instance", "bottomup", "multi_class_bottomup", "multi_class_topdown")


class _Opaque:
    """Stand-in for classes the pickle references but this environment lacks (omegaconf, lightning...)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            return type(name, (_Opaque,), {})


def _pickle_module():
    pm = types.ModuleType("pickle")
    pm.Unpickler = _TolerantUnpickler
    pm.load = lambda f, **k: _TolerantUnpickler(f, **k).load()
    pm.__name__ = "pickle"
    return pm


def load_lightning_state_dict(ckpt_path: str) -> Dict[str, torch.Tensor]:
    """``state_dict`` of a Lightning checkpoint; tensors stay on the CPU."""
    ck = torch.load(ckpt_path, map_location="cpu", weights_only=False, pickle_module=_pickle_module())
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}


@dataclass
class LoadedAssets:
    """What ``load_model_assets`` of the reference returns, in plain-Python form."""

    model_type: str
    backbone_type: str
    backbone_config: dict
    head_config: dict
    preprocessing: dict
    state_dict: Dict[str, torch.Tensor]
    model_dir: str
    node_names: List[str]
    edges: List[Tuple[str, str]]

    def build_model(self):
        from sleap_nn_amd.architectures.model import Model

        m = Model(self.backbone_type, self.backbone_config, self.head_config, self.model_type)
        m.load_state_dict(self.state_dict, strict=True)
        return m


def load_model_assets(model_dir: str, ckpt_name: str = "best.ckpt") -> LoadedAssets:
    cfg_path = os.path.join(model_dir, "training_config.yaml")
    if not os.path.exists(cfg_path):
        raise FileNotFoundError(f"{cfg_path} not found")
    cfg = yaml.safe_load(open(cfg_path))
    mc = cfg["model_config"]
    heads = mc["head_configs"]
    model_type = next((k for k in MODEL_TYPES if heads.get(k)), None)
    if model_type is None:
        raise ValueError(f"no known head config in {cfg_path}: {list(heads)}")
    bb = mc["backbone_config"]
    backbone_type = next((k for k in ("unet", "convnext", "swint") if bb.get(k)), None)
    if backbone_type is None:
        raise ValueError(f"no backbone config in {cfg_path}")
    skels = cfg.get("data_config", {}).get("skeletons") or []
    nodes = [n["name"] for n in skels[0]["nodes"]] if skels else []
    edges = [(e["source"]["name"], e["destination"]["name"]) for e in skels[0].get("edges", [])] if skels else []
    sd = load_lightning_state_dict(os.path.join(model_dir, ckpt_name))
    return LoadedAssets(model_type, backbone_type, bb[backbone_type], heads[model_type], cfg.get("data_config", {}).get("preprocessing", {}) or {},
                        sd, model_dir, nodes, edges)
