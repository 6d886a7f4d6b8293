"""Shared helpers of the model restatements (test infrastructure)."""
from collections import OrderedDict

import numpy as np
import torch

from . import ops

LN_EPS = {"layer_norm": 1e-5, "layer_norm_eps_1e-6": 1e-6}     # layers/factory.py:42-50
BN_EPS = {"batch_norm": 1e-5, "batch_norm_tf": 1e-3}            # layers/factory.py:22-37


class W:
    """Weight accessor: tfimm names -> torch fp32 tensors."""

    def __init__(self, weights):
        self.w = weights

    def __call__(self, name):
        return ops.as_t(self.w[name])

    def has(self, name):
        return name in self.w

    def ln(self, x, prefix, eps):
        return ops.layer_norm(x, self(prefix + "/gamma"), self(prefix + "/beta"), eps)

    def bn(self, x, prefix, eps):
        return ops.batch_norm(x, self(prefix + "/gamma"), self(prefix + "/beta"),
                              self(prefix + "/moving_mean"), self(prefix + "/moving_variance"), eps)

    def dense(self, x, prefix, bias=True):
        return ops.dense(x, self(prefix + "/kernel"), self(prefix + "/bias") if bias else None)


def finish(x, features, return_features):
    y = x.detach().numpy()
    if not return_features:
        return y
    return y, OrderedDict((k, v.detach().numpy()) for k, v in features.items())


def mlp(w: W, x, prefix, act):
    """layers/transformers.py:208-214 MLP.call: fc1 -> act -> (drop) -> fc2 -> (drop)."""
    x = w.dense(x, prefix + "/fc1")
    x = ops.activation(x, act)
    return w.dense(x, prefix + "/fc2")
