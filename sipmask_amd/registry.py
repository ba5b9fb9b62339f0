"""Plugin seam of the hot path: name -> class tables and config-dict construction.

Behavioural contract taken from the reference (M/mmdet/utils/registry.py:7-79,
M/mmdet/models/registry.py:1-9, M/mmdet/models/builder.py:8-43), written without mmcv:
  * ``@HEADS.register_module`` (bare or with ``force=True``) files a class under its __name__;
    a second registration of the same name is a KeyError unless forced; non-classes are a TypeError;
  * ``build_from_cfg(cfg, registry, default_args)`` pops ``type`` (a registered name or a class),
    fills missing keys from ``default_args`` and calls the class with the rest;
  * ``build_detector(cfg, train_cfg, test_cfg)`` passes the two cfgs as default args.
"""
import inspect

from torch import nn


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def __repr__(self):
        return "Registry(name=%s, items=%s)" % (self.name, sorted(self.module_dict))

    def get(self, key):
        return self.module_dict.get(key)

    def register_module(self, cls=None, force=False):
        def _file(klass):
            if not inspect.isclass(klass):
                raise TypeError("module must be a class, but got %s" % type(klass))
            if klass.__name__ in self.module_dict and not force:
                raise KeyError("%s is already registered in %s" % (klass.__name__, self.name))
            self.module_dict[klass.__name__] = klass
            return klass

        return _file if cls is None else _file(cls)


def build_from_cfg(cfg, registry, default_args=None):
    if not (isinstance(cfg, dict) and "type" in cfg):
        raise AssertionError("cfg must be a dict with a 'type' key")
    if default_args is not None and not isinstance(default_args, dict):
        raise AssertionError("default_args must be a dict or None")
    kwargs = dict(cfg)
    kind = kwargs.pop("type")
    if isinstance(kind, str):
        klass = registry.get(kind)
        if klass is None:
            raise KeyError("%s is not in the %s registry" % (kind, registry.name))
    elif inspect.isclass(kind):
        klass = kind
    else:
        raise TypeError("type must be a str or valid type, but got %s" % type(kind))
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return klass(**kwargs)


BACKBONES, NECKS, HEADS, LOSSES, DETECTORS = (Registry(n) for n in ("backbone", "neck", "head", "loss", "detector"))


def _build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return _build(cfg, BACKBONES)


def build_neck(cfg):
    return _build(cfg, NECKS)


def build_head(cfg):
    return _build(cfg, HEADS)


def build_loss(cfg):
    return _build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return _build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
