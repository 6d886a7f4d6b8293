"""``create_model`` / ``create_preprocessing`` / ``transfer_weights``.

Same call signatures, keyword semantics and error behaviour as the reference's
tfimm/models/factory.py (create_model :18-125, create_preprocessing :128-171,
transfer_weights :174-250, _transform_first_conv :282-305); models are engine objects
(:class:`tfimm.models.model.Model`) instead of ``tf.keras.Model``.

Differences forced by the environment (documented in INTEGRATION.md):
  * ``model_path`` / the model cache hold ``.npz`` weight files written by
    ``model.save_weights`` (the reference stores Keras SavedModels).
  * ``pretrained=True`` with a ``[timm]`` / ``[pytorch]`` / ``[hf-pytorch]`` url needs
    network access + timm and raises ``NotImplementedError`` when the weights are not cached.
"""
import logging
import re
from copy import deepcopy
from typing import Callable, List, Optional

import numpy as np

from ..utils.cache import cached_model_path
from .registry import is_model, model_class, model_config


def create_model(
    model_name: str,
    pretrained: bool = False,
    model_path: str = "",
    *,
    in_channels: Optional[int] = None,
    nb_classes: Optional[int] = None,
    **kwargs,
):
    if not is_model(model_name):
        raise RuntimeError(f"Unknown model {model_name}.")
    cls = model_class(model_name)
    cfg = model_config(model_name)

    loaded_model = None
    if model_path or pretrained:
        path = model_path or cached_model_path(model_name)
        if not path:
            if cfg.url.startswith(("[timm]", "[pytorch]", "[hf-pytorch]")):
                raise NotImplementedError(
                    f"Pretrained weights for {model_name} ({cfg.url}) must be converted with timm/torch "
                    "and network access; neither is available. Put an .npz in the model cache instead.")
            raise NotImplementedError(
                "Model not found in cache. Download of weights only implemented for PyTorch models.")
        loaded_model = cls(deepcopy(cfg))
        loaded_model.load_weights(path)

    # config overrides: unknown keys are warned about and ignored (factory.py:92-103)
    cfg = deepcopy(cfg)
    for key, value in kwargs.items():
        if hasattr(cfg, key):
            setattr(cfg, key, value)
        else:
            logging.warning(f"Config for {model_name} does not have field `{key}`. Ignoring field.")
    if in_channels is not None:
        setattr(cfg, "in_channels", in_channels)
    if nb_classes is not None:
        setattr(cfg, "nb_classes", nb_classes)
    model_kwargs = {}
    if "name" in kwargs:
        model_kwargs["name"] = kwargs["name"]

    if loaded_model is not None and loaded_model.cfg == cfg:
        return loaded_model
    model = cls(cfg, **model_kwargs)
    if loaded_model is not None:
        transfer_weights(loaded_model, model)
    return model


def create_preprocessing(model_name: str, *, in_channels: Optional[int] = None,
                         dtype: Optional[str] = None, defer: bool = False) -> Callable:
    """Function mapping [0, 255] images to model inputs: ``(img / 255 - mean) / std`` with
    mean/std tiled to ``in_channels`` (factory.py:153-169).  Works on numpy arrays and torch
    tensors, single images and batches; returns the input's array type.

    ``defer=True`` (not in the reference): a **uint8** image is not converted on the host but wrapped
    in a ``DeferredInput``; ``model(pre(img))`` then evaluates the same three float32 operations inside
    the kernel that converts the input to the engine's layout -- same bits as the host path, a quarter of
    the input bytes.  Anything that is not uint8 is preprocessed immediately as without the flag."""
    if not is_model(model_name):
        raise ValueError(f"Unknown model: {model_name}.")
    cfg = model_config(model_name)
    out_dtype = np.dtype(dtype or "float32")
    n = in_channels or cfg.in_channels

    def _adapt(v):
        v = np.asarray(v, dtype=np.float64)
        reps = n // v.shape[0] + 1
        return np.tile(v, reps)[:n]

    mean, std = _adapt(cfg.mean), _adapt(cfg.std)

    def _preprocess(img):
        if defer and getattr(img, "dtype", None) is not None and str(img.dtype).endswith("uint8"):
            from .model import DeferredInput
            return DeferredInput(img, mean.astype(np.float32), std.astype(np.float32))
        try:
            import torch
            if isinstance(img, torch.Tensor):
                tdt = {"float16": torch.float16, "float32": torch.float32, "float64": torch.float64,
                       "bfloat16": torch.bfloat16}[str(dtype or "float32")]
                m = torch.as_tensor(mean, dtype=tdt, device=img.device)
                s = torch.as_tensor(std, dtype=tdt, device=img.device)
                return (img.to(tdt) / 255.0 - m) / s
        except ImportError:  # pragma: no cover
            pass
        x = np.asarray(img).astype(out_dtype) / out_dtype.type(255.0)
        return ((x - mean.astype(out_dtype)) / std.astype(out_dtype)).astype(out_dtype)

    return _preprocess


def _layer_name(w_name: str) -> str:
    """"remove/fc/kernel" -> "fc" (factory.py:253-266: drop auxiliary levels and the leaf)."""
    name = ("/" + w_name).replace("/remove/", "/")[1:]
    return name.rsplit("/", 1)[0] if "/" in name else name


def _transform_first_conv(weight: np.ndarray, in_channels: int) -> np.ndarray:
    """factory.py:282-305: sum over RGB for 1 channel, otherwise tile and rescale."""
    if weight.ndim != 4:
        return weight
    src = weight.shape[2]
    if in_channels == src:
        return weight
    if in_channels == 1:
        return weight.sum(axis=2, keepdims=True)
    reps = in_channels // src + 1
    w = np.tile(weight, (1, 1, reps, 1))[:, :, :in_channels, :]
    return w * np.float32(src / in_channels)


def transfer_weights(src_model, dst_model, weights_to_ignore: Optional[List[str]] = None):
    """Copy weights ``src_model`` -> ``dst_model`` (in place), adapting the first conv to
    ``dst.cfg.in_channels``, dropping the classifier when ``nb_classes`` differs, applying
    per-model transforms (e.g. pos-embed resize) and skipping ``weights_to_ignore`` patterns
    (factory.py:174-250)."""
    weights_to_ignore = list(weights_to_ignore or [])
    dst_first_conv = getattr(dst_model.cfg, "first_conv", None)
    if hasattr(src_model.cfg, "nb_classes") and hasattr(dst_model.cfg, "nb_classes"):
        keep_classifier = src_model.cfg.nb_classes == dst_model.cfg.nb_classes
    else:
        keep_classifier = True
    dst_classifier = getattr(dst_model.cfg, "classifier", [])
    if isinstance(dst_classifier, str):
        dst_classifier = [dst_classifier]
    transforms = getattr(src_model, "transform_weights", dict())
    weights_to_ignore += list(getattr(dst_model, "keys_to_ignore_on_load_missing", []))

    src = src_model.weights
    new = {}
    for w_name in dst_model.weights:
        layer = _layer_name(w_name)
        if any(re.search(pat, w_name) is not None for pat in weights_to_ignore):
            continue
        if layer in dst_classifier:
            if keep_classifier:
                new[w_name] = src[w_name]
        elif layer == dst_first_conv:
            new[w_name] = _transform_first_conv(src[w_name], dst_model.cfg.in_channels)
        elif w_name in transforms:
            new[w_name] = transforms[w_name](src_model, src[w_name], dst_model.cfg)
        else:
            new[w_name] = src[w_name]
    dst_model.set_weights(new, strict=False)
