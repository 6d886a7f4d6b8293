"""Model cache directory handling.

Same semantics as reference tfimm/utils/cache.py:11-94: ``$TFIMM_HOME`` overrides
``$XDG_CACHE_HOME/tfimm`` overrides ``~/.cache/tfimm``; a per-model override table is
consulted first.
"""
import os
from typing import List, Optional

_explicit_dir: Optional[str] = None
_per_model = {}


def get_dir() -> str:
    if _explicit_dir is not None:
        return _explicit_dir
    base = os.getenv("XDG_CACHE_HOME", "~/.cache")
    home = os.getenv("TFIMM_HOME", os.path.join(base, "tfimm"))
    return os.path.expanduser(home)


def set_dir(d: str):
    global _explicit_dir
    _explicit_dir = d


def set_model_cache(model_name: str, model_path: str):
    _per_model[model_name] = model_path


def clear_model_cache(model_name: str):
    _per_model.pop(model_name, None)


def list_cached_models() -> List[str]:
    return sorted(_per_model)


def cached_model_path(model_name: str) -> Optional[str]:
    if model_name in _per_model:
        return _per_model[model_name]
    for cand in (os.path.join(get_dir(), model_name),
                 os.path.join(get_dir(), model_name + ".npz")):
        if os.path.exists(cand):
            return cand
    return None
