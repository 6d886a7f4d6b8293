"""Model registry: name -> (model class, config).

API-compatible with reference tfimm/models/registry.py:34-151 (``register_model``,
``list_models`` with fnmatch filters and natural sort, ``model_class``,
``model_config``, ``is_model``, ``is_model_pretrained``, ``list_modules``,
``is_model_in_modules``). Pure Python; importing ``tfimm`` fills it as a side effect.
"""
import fnmatch
import re
import sys
from copy import deepcopy
from typing import Dict, List, Set, Union

__all__ = [
    "list_models", "is_model", "is_model_in_modules", "is_model_pretrained",
    "list_modules", "model_class", "model_config", "register_model",
]


class _Entry:
    __slots__ = ("cls", "cfg", "module")

    def __init__(self, cls, cfg, module):
        self.cls, self.cfg, self.module = cls, cfg, module


_entries: Dict[str, _Entry] = {}


def register_model(fn):
    """Decorator for functions returning ``(cls, cfg)``; ``fn.__name__`` must equal
    ``cfg.name`` (ValueError otherwise, reference registry.py:38-39)."""
    cls, cfg = fn()
    if fn.__name__ != cfg.name:
        raise ValueError(f"Model name({cfg.name}) != function name ({fn.__name__}).")
    module = fn.__module__.split(".")[-1]
    mod = sys.modules.get(fn.__module__)
    if mod is not None:
        if not hasattr(mod, "__all__"):
            mod.__all__ = []
        mod.__all__.append(cfg.name)
    _entries[cfg.name] = _Entry(cls, deepcopy(cfg), module)
    return fn


def _natural_key(s: str):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s.lower())]


def _as_list(x):
    return list(x) if isinstance(x, (tuple, list)) else [x]


def list_models(
    name_filter: Union[str, List[str]] = "",
    module: str = "",
    pretrained: Union[bool, str] = False,
    exclude_filters: Union[str, List[str]] = "",
) -> List[str]:
    names = [n for n, e in _entries.items() if not module or e.module == module]
    if name_filter:
        keep: Set[str] = set()
        for pat in _as_list(name_filter):
            keep.update(fnmatch.filter(names, pat))
    else:
        keep = set(names)
    if exclude_filters:
        for pat in _as_list(exclude_filters):
            keep.difference_update(fnmatch.filter(keep, pat))
    if pretrained is True:
        keep = {n for n in keep if _entries[n].cfg.url}
    elif pretrained == "timm":
        raise NotImplementedError("timm is not available in this environment.")
    return sorted(keep, key=_natural_key)


def is_model(model_name: str) -> bool:
    return model_name in _entries


def model_class(model_name: str):
    return _entries[model_name].cls


def model_config(model_name: str):
    return _entries[model_name].cfg


def list_modules() -> List[str]:
    return sorted({e.module for e in _entries.values()})


def is_model_in_modules(model_name: str, module_names) -> bool:
    assert isinstance(module_names, (tuple, list, set))
    e = _entries.get(model_name)
    return e is not None and e.module in module_names


def is_model_pretrained(model_name: str) -> bool:
    return bool(_entries[model_name].cfg.url)
