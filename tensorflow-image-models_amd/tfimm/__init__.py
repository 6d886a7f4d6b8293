"""tfimm on MI355X: the ``tfimm`` forward-path API backed by hand-written HIP kernels.

Drop-in for the inference path of martinsbruveris/tensorflow-image-models:
``tfimm.create_model(name)(x)``, ``tfimm.list_models()``, ``tfimm.create_preprocessing()``
(reference tfimm/__init__.py:1-12).  Importing the package registers all architectures.
"""
from . import architectures, layers  # noqa: F401
from .models.factory import create_model, create_preprocessing  # noqa: F401
from .models.registry import list_models  # noqa: F401
from .utils import (  # noqa: F401
    cached_model_path,
    clear_model_cache,
    get_dir,
    list_cached_models,
    set_dir,
    set_model_cache,
)

__version__ = "0.1.0+mi355x"
