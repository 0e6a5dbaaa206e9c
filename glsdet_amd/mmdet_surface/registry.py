"""Minimal stand-ins for the two mmcv pieces the reference's model surface rests on
(mmcv is not installable here): `Registry` (`build(cfg)` pops `type`, looks the class up by
name, merges default_args without overriding; ufp/mmdet/models/builder.py:4-59) and
`Config.fromfile` for python-file configs with `_base_` inheritance and `_delete_`
(SURVEY.md Appendix C)."""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._modules: Dict[str, type] = {}

    def register_module(self, name: Optional[str] = None, module: Optional[type] = None, force: bool = False):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError("%s is already registered in %s" % (key, self.name))
            self._modules[key] = cls
            return cls
        return _reg(module) if module is not None else _reg

    def get(self, key: str):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def build(self, cfg, default_args: Optional[dict] = None):
        if not isinstance(cfg, dict):
            raise TypeError("cfg must be a dict, but got %s" % type(cfg))
        if "type" not in cfg and not (default_args and "type" in default_args):
            raise KeyError('`cfg` or `default_args` must contain the key "type", but got %s' % (cfg,))
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        typ = args.pop("type")
        cls = typ if isinstance(typ, type) else self.get(typ)
        if cls is None:
            raise KeyError("%s is not in the %s registry" % (typ, self.name))
        return cls(**args)


# all model kinds alias ONE registry, as in ufp/mmdet/models/builder.py:7-15
MODELS = Registry("models")
BACKBONES = NECKS = HEADS = DETECTORS = MODELS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """ufp/mmdet/models/builder.py:42-59"""
    assert cfg.get("train_cfg") is None or train_cfg is None, "train_cfg specified in both outer field and model field"
    assert cfg.get("test_cfg") is None or test_cfg is None, "test_cfg specified in both outer field and model field"
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


class ConfigDict(dict):
    """dict with attribute access (cfg.model.backbone.type)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(base: dict, new: dict) -> dict:
    out = copy.deepcopy(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != "_delete_"}
            out[k] = copy.deepcopy(v)
    return out


class Config:
    def __init__(self, cfg_dict: dict, filename: Optional[str] = None):
        object.__setattr__(self, "_cfg", _wrap(cfg_dict))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def _load(path: str) -> dict:
        path = os.path.abspath(path)
        scope: Dict[str, Any] = {"__file__": path}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), scope)
        cfg = {k: v for k, v in scope.items() if not k.startswith("__") and not callable(v)
               and type(v).__name__ != "module"}
        bases = cfg.pop("_base_", [])
        if isinstance(bases, str):
            bases = [bases]
        merged: dict = {}
        for b in bases:
            merged = _merge(merged, Config._load(os.path.join(os.path.dirname(path), b)))
        return _merge(merged, cfg)

    @staticmethod
    def fromfile(path: str) -> "Config":
        return Config(Config._load(path), path)

    def merge_from_dict(self, options: dict):
        cur = dict(self._cfg)
        for key, v in options.items():
            d = cur
            parts = key.split(".")
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, "_cfg", _wrap(cur))

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __getitem__(self, k):
        return self._cfg[k]

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def __contains__(self, k):
        return k in self._cfg
