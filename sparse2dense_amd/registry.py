"""Registry + builders with the reference's key names (`det3d/models/registry.py:3-10`,
`det3d/utils/registry.py:6-78`, `det3d/models/builder.py:16-50`): configs say
`dict(type="SpMiddleResNetFHD", ...)` and `build_detector(cfg.S_model)` returns our modules."""
import copy
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return f"Registry(name={self._name}, items={list(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        if cls.__name__ in self._module_dict:
            raise KeyError(f"{cls.__name__} is already registered in {self._name}")
        self._module_dict[cls.__name__] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    """cfg['type'] names a registered class; remaining keys (+defaults) are constructor kwargs."""
    if not (isinstance(cfg, dict) and "type" in cfg):
        raise TypeError("cfg must be a dict with a 'type' key")
    args = copy.deepcopy(dict(cfg))
    kind = args.pop("type")
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f"{kind} is not in the {registry.name} registry")
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(kind)}")
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    return cls(**args)


READERS = Registry("reader")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
SECOND_STAGE = Registry("second_stage")
ROI_HEAD = Registry("roi_head")


def _ensure_registered():
    """The model modules register themselves on import; make `build_*` usable on its own."""
    from . import backbones, detectors, heads, necks, pillars, second_stage  # noqa: F401


def build(cfg, registry, default_args=None):
    _ensure_registered()
    if isinstance(cfg, (list, tuple)):
        from torch import nn
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_reader(cfg):
    return build(cfg, READERS)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
