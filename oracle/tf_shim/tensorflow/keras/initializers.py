"""``tf.keras.initializers``: distributions only matter for ``create_model``'s freshly built
weights (parity tests overwrite every variable), so the random ones draw from one numpy
generator; shapes, fans and constants follow Keras."""
import math

import numpy as np

from .._core import Tensor, _raw, as_dtype

_rng = np.random.default_rng(0)


def seed(s):
    global _rng
    _rng = np.random.default_rng(s)


def _out(a, dtype):
    return Tensor(_raw(np.asarray(a), dtype or "float32"))


def _fans(shape):
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


class Initializer:
    def __call__(self, shape, dtype=None, **kwargs):
        raise NotImplementedError

    def get_config(self):
        return {}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


class Zeros(Initializer):
    def __call__(self, shape, dtype=None, **kwargs):
        return _out(np.zeros(shape), dtype)


class Ones(Initializer):
    def __call__(self, shape, dtype=None, **kwargs):
        return _out(np.ones(shape), dtype)


class Constant(Initializer):
    def __init__(self, value=0):
        self.value = value

    def __call__(self, shape, dtype=None, **kwargs):
        v = np.asarray(self.value.numpy() if hasattr(self.value, "numpy") else self.value)
        return _out(np.broadcast_to(v, shape).copy() if v.ndim == 0 else v.reshape(shape), dtype)


class RandomNormal(Initializer):
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, dtype=None, **kwargs):
        return _out(_rng.normal(self.mean, self.stddev, size=shape), dtype)


class TruncatedNormal(Initializer):
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, dtype=None, **kwargs):
        a = _rng.normal(0.0, 1.0, size=shape)
        while True:
            bad = np.abs(a) > 2.0
            if not bad.any():
                break
            a[bad] = _rng.normal(0.0, 1.0, size=int(bad.sum()))
        return _out(self.mean + self.stddev * a, dtype)


class RandomUniform(Initializer):
    def __init__(self, minval=-0.05, maxval=0.05, seed=None):
        self.minval, self.maxval = minval, maxval

    def __call__(self, shape, dtype=None, **kwargs):
        return _out(_rng.uniform(self.minval, self.maxval, size=shape), dtype)


class VarianceScaling(Initializer):
    def __init__(self, scale=1.0, mode="fan_in", distribution="truncated_normal", seed=None):
        self.scale, self.mode, self.distribution = scale, mode, distribution

    def __call__(self, shape, dtype=None, **kwargs):
        fi, fo = _fans(tuple(shape))
        n = {"fan_in": fi, "fan_out": fo, "fan_avg": (fi + fo) / 2.0}[self.mode]
        s = self.scale / max(1.0, n)
        if self.distribution == "uniform":
            lim = math.sqrt(3.0 * s)
            return _out(_rng.uniform(-lim, lim, size=shape), dtype)
        if self.distribution == "truncated_normal":
            return TruncatedNormal(0.0, math.sqrt(s) / 0.87962566103423978)(shape, dtype)
        return _out(_rng.normal(0.0, math.sqrt(s), size=shape), dtype)


class GlorotUniform(VarianceScaling):
    def __init__(self, seed=None):
        super().__init__(1.0, "fan_avg", "uniform")


class GlorotNormal(VarianceScaling):
    def __init__(self, seed=None):
        super().__init__(1.0, "fan_avg", "truncated_normal")


class HeNormal(VarianceScaling):
    def __init__(self, seed=None):
        super().__init__(2.0, "fan_in", "truncated_normal")


class HeUniform(VarianceScaling):
    def __init__(self, seed=None):
        super().__init__(2.0, "fan_in", "uniform")


class LecunNormal(VarianceScaling):
    def __init__(self, seed=None):
        super().__init__(1.0, "fan_in", "truncated_normal")


constant = Constant
zeros = Zeros
ones = Ones

_BY_NAME = {"zeros": Zeros, "ones": Ones, "glorot_uniform": GlorotUniform, "glorot_normal": GlorotNormal,
            "he_normal": HeNormal, "he_uniform": HeUniform, "lecun_normal": LecunNormal,
            "random_normal": RandomNormal, "truncated_normal": TruncatedNormal, "random_uniform": RandomUniform,
            "constant": Constant}


def get(identifier):
    if identifier is None:
        return None
    if isinstance(identifier, str):
        if identifier not in _BY_NAME:
            raise ValueError(f"Unknown initializer: {identifier}")
        return _BY_NAME[identifier]()
    if isinstance(identifier, type):
        return identifier()
    if callable(identifier):
        return identifier
    raise ValueError(f"Could not interpret initializer identifier: {identifier}")


__all__ = ["as_dtype"]
