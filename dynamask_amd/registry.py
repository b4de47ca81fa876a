"""mmdet-style plugin boundary: registries + config loader.

Mirrors the reference's ``mmdet/models/builder.py:5-63`` (HEADS /
ROI_EXTRACTORS / LOSSES registries, ``build_from_cfg`` popping ``type`` and
calling ``cls(**cfg)``) and mmcv's python-dict ``Config.fromfile`` with
``_base_`` inheritance (``train.py:67-69`` of the reference), so that
``configs/dynamask/coco/r50-dynamask-1x.py`` is consumed unchanged.
"""
import os

import torch.nn as nn


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def __contains__(self, key):
        return key in self._module_dict

    def _register(self, cls, name=None, force=False):
        name = name or cls.__name__
        if not force and name in self._module_dict:
            raise KeyError(f'{name} is already registered in {self._name}')
        self._module_dict[name] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def _reg(cls):
            self._register(cls, name, force)
            return cls
        return _reg


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        raise KeyError(f'`cfg` must contain the key "type", but got {cfg}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif isinstance(obj_type, type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    return obj_cls(**args)


ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
ROI_LAYERS = Registry('roi_layer')       # stands in for getattr(mmcv.ops, type) (base_roi_extractor.py:49-55)
UPSAMPLE_LAYERS = Registry('upsample')   # stands in for mmcv's build_upsample_layer


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_roi_extractor(cfg):
    return build(cfg, ROI_EXTRACTORS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


class ConfigDict(dict):
    """dict with attribute access (train_cfg.flops etc.)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _to_cfgdict(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _to_cfgdict(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cfgdict(v) for v in obj)
    return obj


def _merge(base, new):
    out = dict(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != '_delete_'}
            out[k] = v
    return out


class Config:
    """Minimal python-dict config loader with ``_base_`` inheritance."""

    @staticmethod
    def _file2dict(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        scope = {}
        with open(filename, 'r') as f:
            exec(compile(f.read(), filename, 'exec'), scope)
        cfg = {k: v for k, v in scope.items() if not k.startswith('__') and not callable(v)
               and not isinstance(v, type(os))}
        base = cfg.pop('_base_', None)
        if base is not None:
            bases = base if isinstance(base, (list, tuple)) else [base]
            merged = {}
            for b in bases:
                merged = _merge(merged, Config._file2dict(os.path.join(os.path.dirname(filename), b)))
            cfg = _merge(merged, cfg)
        return cfg

    @staticmethod
    def fromfile(filename):
        return _to_cfgdict(Config._file2dict(filename))
