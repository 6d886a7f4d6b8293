"""Weight initialisers for models built without pretrained weights.

``mode="keras"`` reproduces the *distributions* Keras would draw for the reference's layers
(glorot-uniform kernels, zero biases, BN gamma=1/beta=0/mean=0/var=1, zero tokens and
position embeddings) including the reference's overrides such as the zero-initialised last
BatchNorm of every ResNet block (resnet.py:249-256).  As SURVEY.md App. B notes, that init
makes whole code paths numerically invisible, so parity tests and benchmarks use
``mode="synthetic"``: a deterministic, non-degenerate generator (He-normal kernels, random
affine/BN statistics, random tokens) drawn in sorted-name order from one
``np.random.default_rng(seed)``.
"""
from collections import OrderedDict
from typing import Dict

import numpy as np


def _fans(shape, kind):
    if kind == "dwconv":       # (kh, kw, C, 1): each output sees kh*kw inputs
        rf = shape[0] * shape[1]
        return rf, rf
    if len(shape) == 4:
        rf = shape[0] * shape[1]
        return shape[2] * rf, shape[3] * rf
    if len(shape) == 2:
        return shape[0], shape[1]
    n = int(np.prod(shape))
    return n, n


def initialize(specs, mode: str = "keras", seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = OrderedDict()
    names = sorted(specs) if mode == "synthetic" else list(specs)
    for name in names:
        spec = specs[name]
        shape, kind = tuple(spec.shape), spec.kind
        if mode == "keras":
            if spec.init == "zeros":
                w = np.zeros(shape)
            elif spec.init == "ones":
                w = np.ones(shape)
            elif kind in ("conv", "dwconv", "dense"):
                fi, fo = _fans(shape, kind)
                lim = np.sqrt(6.0 / (fi + fo))
                w = rng.uniform(-lim, lim, size=shape)
            elif kind in ("gamma", "var"):
                w = np.ones(shape)      # also for init == "small" (a hint for the synthetic mode only)
            elif kind == "scale":   # LayerScale: filled with the config's init value
                w = np.full(shape, float(spec.init or 1e-4))
            else:                   # bias, beta, mean, token, pos, table
                w = np.zeros(shape)
        elif mode == "synthetic":
            if kind in ("conv", "dwconv", "dense"):
                fi, _ = _fans(shape, kind)
                w = rng.normal(0.0, np.sqrt(2.0 / fi), size=shape)
            elif kind == "gamma" and spec.init in ("zeros", "small"):
                # last BN of a residual branch (zero-initialised in the reference): keep the
                # branch visible but small so 16-50 stacked blocks stay well conditioned
                w = rng.uniform(0.1, 0.3, size=shape)
            elif kind in ("gamma", "var", "scale"):
                w = rng.uniform(0.5, 1.5, size=shape)
            elif kind == "mean":
                w = rng.normal(0.0, 0.1, size=shape)
            elif kind in ("token", "pos", "table"):
                w = rng.normal(0.0, 0.1, size=shape)
            else:                   # bias, beta
                w = rng.normal(0.0, 0.05, size=shape)
        else:
            raise ValueError(f"unknown init mode {mode}")
        out[name] = w.astype(np.float32)
    # keep the spec's declaration order for iteration
    return OrderedDict((k, out[k]) for k in specs)


def synthetic_weights(model, seed: int = 2021) -> Dict[str, np.ndarray]:
    """Non-degenerate deterministic weights for ``model`` (parity tests, benchmarks)."""
    return initialize(model._specs, mode="synthetic", seed=seed)
