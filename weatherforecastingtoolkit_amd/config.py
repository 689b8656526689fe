"""YAML + `key=value` dotlist overrides with the semantics the reference gets from
OmegaConf (experiments/ae_v2/train.py:264-277) — OmegaConf is not a dependency."""
from __future__ import annotations

import yaml


class Cfg(dict):
    """dict with attribute access (cfg.optim.lr), like an OmegaConf node."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return Cfg({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def _fix_scalars(o):
    # PyYAML reads '5e-5' as a string (YAML 1.1); OmegaConf reads it as a float
    if isinstance(o, dict):
        return {k: _fix_scalars(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_fix_scalars(v) for v in o]
    if isinstance(o, str):
        try:
            return float(o) if any(c in o for c in ".eE") else int(o)
        except ValueError:
            return o
    return o


def load(path, carried=None):
    """the YAML file, over `carried` — keys of the reference's config that this build accepts (so that existing
    override command lines keep parsing) but does not read: they live in code instead of the YAML"""
    with open(path) as f:
        cfg = _wrap(_fix_scalars(yaml.safe_load(f)))
    if carried:
        cfg = merge(_wrap(_fix_scalars(_deep_copy(carried))), cfg)
    return cfg


def _deep_copy(o):
    if isinstance(o, dict):
        return {k: _deep_copy(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_deep_copy(v) for v in o]
    return o


def from_dotlist(items):
    out = {}
    for it in items:
        if "=" not in it:
            raise ValueError(f"override '{it}' is not key=value")
        k, v = it.split("=", 1)
        node = out
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _fix_scalars(yaml.safe_load(v))
    return out


def merge(base, over):
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            merge(base[k], v)
        else:
            base[k] = _wrap(v)
    return base
