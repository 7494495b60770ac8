"""Small host-side helpers shared by the package."""
from __future__ import annotations


def cfg_get(cfg, key, default=None):
    """Read ``key`` from a dict / attribute-style config (DictConfig, attrs, namespace)."""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    try:
        return cfg[key]
    except Exception:
        return getattr(cfg, key, default)


def cfg_keys(cfg):
    if isinstance(cfg, dict):
        return list(cfg.keys())
    try:
        return list(cfg.keys())
    except Exception:
        return [k for k in vars(cfg) if not k.startswith("_")]


def to_plain(cfg):
    """Recursively convert a config object to plain dict / list / scalars."""
    if isinstance(cfg, dict) or hasattr(cfg, "keys"):
        return {k: to_plain(cfg_get(cfg, k)) for k in cfg_keys(cfg)}
    if isinstance(cfg, (list, tuple)):
        return [to_plain(v) for v in cfg]
    return cfg
