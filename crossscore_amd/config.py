"""Minimal Hydra/OmegaConf-compatible config tree for the CrossScore hot path.

Hydra and OmegaConf are not installed on the GPU box, so this module composes the same key tree the reference
builds with `@hydra.main(config_path="../config", config_name="default_predict")` (task/predict.py:21):
a `defaults:` list of `group: name` entries merged under the group key, then `a.b=c` command-line overrides.
Attribute and item access both work (`cfg.model.patch_size`, `cfg["model"]["patch_size"]`), which is all the
model code uses (task/core.py:39-56, model/cross_reference.py:20-42).
"""
from __future__ import annotations

import copy
import os
from typing import Any, Iterable, Optional

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")


class Cfg(dict):
    """dict with attribute access, recursively."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v: Any) -> Any:
    if isinstance(v, dict) and not isinstance(v, Cfg):
        return Cfg({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _parse_scalar(s: str) -> Any:
    return yaml.safe_load(s) if s != "" else ""


def apply_overrides(cfg: Cfg, overrides: Iterable[str]) -> Cfg:
    """Hydra-style `a.b.c=value` overrides (value parsed as YAML: ints, floats, bools, null, lists)."""
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        key, val = ov.split("=", 1)
        key = key.lstrip("+")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in node or not isinstance(node[p], dict):
                node[p] = Cfg()
            node = node[p]
        node[parts[-1]] = _wrap(_parse_scalar(val))
    return cfg


def load_config(name: str = "default_predict", overrides: Optional[Iterable[str]] = None,
                config_dir: str = CONFIG_DIR) -> Cfg:
    """Compose `<config_dir>/<name>.yaml` with its defaults list and apply overrides."""
    with open(os.path.join(config_dir, name + ".yaml")) as f:
        root = yaml.safe_load(f) or {}
    defaults = root.pop("defaults", [])
    out: dict = {}
    self_merged = False
    for d in defaults:
        if d == "_self_":
            _merge(out, root)
            self_merged = True
        elif isinstance(d, dict):
            for group, choice in d.items():
                if group.startswith("override "):
                    continue  # hydra/* logging overrides: nothing to compose
                with open(os.path.join(config_dir, group, f"{choice}.yaml")) as f:
                    sub = yaml.safe_load(f) or {}
                _merge(out.setdefault(group, {}), sub)
    if not self_merged:
        _merge(out, root)
    cfg = _wrap(out)
    if overrides:
        apply_overrides(cfg, overrides)
    return cfg


def model_config(**over) -> Cfg:
    """Just the `model` group wrapped as cfg.model, with keyword overrides on dotted paths
    (e.g. model_config(**{"backbone.from_pretrained": "facebook/dinov2-base"}))."""
    with open(os.path.join(CONFIG_DIR, "model", "model.yaml")) as f:
        cfg = _wrap({"model": yaml.safe_load(f)})
    for k, v in over.items():
        node = cfg.model
        parts = k.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = _wrap(v)
    return cfg
