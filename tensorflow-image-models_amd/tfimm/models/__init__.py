from .config import ModelConfig  # noqa: F401
from .factory import create_model, create_preprocessing, transfer_weights  # noqa: F401
from .model import DeferredInput, Model, Tensor, WeightSpec  # noqa: F401
from .registry import (  # noqa: F401
    is_model,
    is_model_in_modules,
    is_model_pretrained,
    list_models,
    list_modules,
    model_class,
    model_config,
    register_model,
)
