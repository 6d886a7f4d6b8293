"""PyTorch / timm ``state_dict`` -> engine weights (reference tfimm/utils/timm.py).

The engine's weights carry the reference's TF variable names, so the reference's naming and layout
rules are the interchange format:

  name   drop ":0" and the first path level (the model name), "/remove/" -> "/", "a___b" -> "b",
         "_._" -> "/"; leaf kernel | depthwise_kernel | embeddings | gamma -> weight, beta -> bias,
         moving_mean -> running_mean, moving_variance -> running_var; "/" -> "."   (timm.py:58-104)
  layout rank-4 kernels: PyTorch OIHW -> HWIO (transpose(2, 3, 1, 0)); other kernels transposed;
         then squeeze / expand_dims / reshape to the variable's shape, which also turns a depthwise
         (C, 1, kh, kw) into (kh, kw, C, 1)                                           (timm.py:164-197)
  state_dict keys ending in ".beta" / ".gamma" (ResMLP affine, ConvNeXt LayerScale) are first
  renamed to ".bias" / ".weight"; "num_batches_tracked" is ignored                  (timm.py:120-135, 205-207)

Downloading checkpoints (timm hub, torch hub, URLs) needs network access and is out of scope:
those entry points raise ``NotImplementedError`` unless the package they need is importable.
"""
import logging
import re
from typing import Dict, Optional, Tuple

import numpy as np

NO, SIMPLE, CONV2D = "no", "simple", "conv2d"


def convert_tf_weight_name_to_pt_weight_name(tf_name: str, tf_weight_shape: Optional[Tuple[int, ...]] = None):
    """TF variable name -> (PyTorch key, transposition kind).  ``tf_weight_shape`` is a plain tuple."""
    name = tf_name.replace(":0", "")
    name = re.sub(r"/[^/]*___([^/]*)/", r"/\1/", name)
    name = name.replace("_._", "/").replace("/remove/", "/")
    name = re.sub(r"//+", "/", name)
    parts = name.split("/")
    if len(parts) > 1:
        parts = parts[1:]
    leaf = parts[-1]
    if leaf in ("kernel", "depthwise_kernel") and tf_weight_shape is not None and len(tf_weight_shape) == 4:
        kind = CONV2D
    elif leaf in ("kernel", "pointwise_kernel", "depthwise_kernel") or "emb_projs" in parts or "out_projs" in parts:
        kind = SIMPLE
    else:
        kind = NO
    rename = {"kernel": "weight", "depthwise_kernel": "weight", "embeddings": "weight", "gamma": "weight",
              "beta": "bias", "moving_mean": "running_mean", "moving_variance": "running_var"}
    parts[-1] = rename.get(leaf, leaf)
    return ".".join(parts), kind


def _to_numpy(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().float().numpy()
    return np.asarray(v)


def load_pytorch_weights_in_model(model, pt_state_dict: Dict[str, object], allow_missing_keys: bool = False):
    """Load a PyTorch ``state_dict`` (tensors or arrays) into an engine model, in place.
    Mirrors ``load_pytorch_weights_in_tf2_model`` (timm.py:109-229): raises ``AttributeError`` for a
    weight the state_dict lacks (unless allowed / ignorable), warns about unused keys."""
    sd = dict(pt_state_dict)
    for key in list(sd):
        if key.endswith(".beta"):
            sd[key[:-len(".beta")] + ".bias"] = sd.pop(key)
        elif key.endswith(".gamma"):
            sd[key[:-len(".gamma")] + ".weight"] = sd.pop(key)
    unused = set(sd)
    missing, new = [], {}
    ignorable = tuple(getattr(model, "keys_to_ignore_on_load_missing", ()))
    for w_name, cur in model.weights.items():
        full = f"{model.name}/{w_name}:0"
        shape = tuple(cur.shape)
        key, kind = convert_tf_weight_name_to_pt_weight_name(full, shape)
        if key not in sd:
            if allow_missing_keys:
                missing.append(key)
                continue
            if any(re.search(pat, full) is not None for pat in ignorable):
                continue
            raise AttributeError(f"{key} not found in PyTorch model")
        arr = _to_numpy(sd[key])
        if kind == CONV2D:
            arr = np.transpose(arr, (2, 3, 1, 0))
        elif kind == SIMPLE:
            arr = np.transpose(arr)
        if len(shape) < arr.ndim:
            arr = np.squeeze(arr)
        elif len(shape) > arr.ndim:
            arr = np.expand_dims(arr, 0)
        if tuple(arr.shape) != shape:
            try:
                arr = np.reshape(arr, shape)
            except ValueError as e:
                e.args += (key, full)
                raise
        new[w_name] = np.ascontiguousarray(arr, dtype=np.float32)
        unused.discard(key)
    model.set_weights(new, strict=False)
    unused = sorted(k for k in unused if "num_batches_tracked" not in k)
    if unused:
        logging.warning(f"Some weights of the PyTorch model were not used when initializing {type(model).__name__}: {unused}.")
    if missing:
        logging.warning(f"Some weights of {type(model).__name__} were not initialized from the PyTorch model: {missing}.")
    return model


# the reference's public name
load_pytorch_weights_in_tf2_model = load_pytorch_weights_in_model


def load_timm_weights(model, model_name: str):
    """timm.py:232-255: ``timm.create_model(model_name, pretrained=True).state_dict()`` -> model."""
    try:
        import timm
    except ImportError as e:
        raise NotImplementedError("timm is not installed; convert the checkpoint elsewhere and use "
                                  "load_pytorch_weights_in_model(model, state_dict) or model.load_weights(npz)") from e
    pt_model = timm.create_model(model_name, pretrained=True)
    return load_pytorch_weights_in_model(model, pt_model.state_dict())


def load_pth_url_weights(model, url: str):
    """timm.py:273-282 downloads a checkpoint: no network in this environment."""
    raise NotImplementedError("downloading checkpoints needs network access; load the file yourself and call "
                              "load_pytorch_weights_in_model(model, state_dict)")
