"""Minimal YAML + ``_target_`` object construction (the reference uses OmegaConf + hydra, both absent offline).

Implements what the reference's configs and ``instantiate_any`` rely on (reference src/duwu/utils/__init__.py:25-50,
test_scripts/test_train.py:23-33): dotted-path targets, ``_partial_``, ``_recursive_`` (default true), ``_args_``,
the custom ``{class, factory, args, kwargs}`` dialect, and deep-merge of several config files.

Targets that name packages which are not installable offline are resolved to this build's local equivalents
(ALIASES); nothing is ever fetched from a hub.
"""
import copy
import functools
import importlib

import yaml

ALIASES = {
    "diffusers.EulerDiscreteScheduler": "uwudiff_amd.scheduler.EulerDiscreteScheduler",
    "transformers.CLIPTextModel": "uwudiff_amd.conditioning.SyntheticCLIPTextModel",  # kind "clip_sd1": normed ctx = LN(layer_idx)
    "transformers.CLIPTextModelWithProjection": "uwudiff_amd.conditioning.SyntheticTextModel",
    "lightning.pytorch.callbacks.ModelCheckpoint": "uwudiff_amd.engine.ModelCheckpoint",
    "lightning.pytorch.callbacks.LearningRateMonitor": "uwudiff_amd.engine.LearningRateMonitor",
}


class AttrDict(dict):
    """dict with attribute access (``config.seed``), enough of DictConfig for the launcher."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return AttrDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def load_yaml(path):
    with open(path) as f:
        return _wrap(yaml.safe_load(f) or {})


def merge(*cfgs):
    def _m(a, b):
        out = AttrDict(a)
        for k, v in b.items():
            out[k] = _m(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else copy.deepcopy(v)
        return out

    res = AttrDict()
    for c in cfgs:
        res = _m(res, _wrap(c))
    return res


def get_obj_from_str(string, reload=False):
    """Resolve a dotted path to an object; walks ``a.b.C.method`` and applies ALIASES for absent packages."""
    for src, dst in ALIASES.items():
        if string == src or string.startswith(src + "."):
            string = dst + string[len(src):]
            break
    parts = string.split(".")
    last_err = None
    for i in range(len(parts), 0, -1):
        modname = ".".join(parts[:i])
        try:
            obj = importlib.import_module(modname)
        except ModuleNotFoundError as e:
            last_err = e
            continue
        if reload:
            importlib.reload(obj)
        for attr in parts[i:]:
            obj = getattr(obj, attr)
        return obj
    raise ModuleNotFoundError(f"cannot resolve target {string!r}: {last_err}")


def _is_node(x):
    return isinstance(x, dict) and "_target_" in x


def instantiate(node, *args, **overrides):
    """hydra.utils.instantiate subset."""
    if not _is_node(node):
        raise ValueError("instantiate() needs a mapping with a _target_")
    node = dict(node)
    target = node.pop("_target_")
    partial = bool(node.pop("_partial_", False))
    recursive = bool(node.pop("_recursive_", True))
    pos = list(node.pop("_args_", []))
    node.pop("_convert_", None)
    node.update(overrides)
    fn = get_obj_from_str(target) if isinstance(target, str) else target

    def build(v):
        if _is_node(v):
            return instantiate(v)
        if isinstance(v, dict):
            return AttrDict({k: build(x) for k, x in v.items()})
        if isinstance(v, list):
            return [build(x) for x in v]
        return v

    if recursive:
        node = {k: build(v) for k, v in node.items()}
        pos = [build(v) for v in pos]
    else:
        node = {k: _wrap(v) for k, v in node.items()}
    pos = list(args) + pos
    if partial:
        return functools.partial(fn, *pos, **node)
    return fn(*pos, **node)


def instantiate_class(obj):
    """The custom ``{class, factory, args, kwargs}`` / dotted-string dialect (utils/__init__.py:25-38)."""
    if isinstance(obj, dict) and "class" in obj:
        obj = dict(obj)
        factory = instantiate_class(obj.pop("class"))
        if "factory" in obj:
            factory = getattr(factory, obj.pop("factory"))
        if "args" in obj or "kwargs" in obj:
            return factory(*obj.get("args", []), **obj.get("kwargs", {}))
        return factory(**obj)
    if isinstance(obj, str):
        return get_obj_from_str(obj)
    return obj


def instantiate_any(obj):
    """Both dialects (utils/__init__.py:41-50)."""
    if _is_node(obj):
        return instantiate(obj)
    return instantiate_class(obj)
