"""A stand-in ``tensorflow`` module on torch-CPU -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Why it exists: the reference (tfimm v0.2.14) is pure Python whose arithmetic lives in
TensorFlow/Keras 2.12 (poetry.lock:1405-1406, 608-609), which cannot be installed in the build
container.  With ``oracle/tf_shim`` in front of ``sys.path``, ``import tensorflow`` resolves to
this package and the reference's OWN model code under /root/reference/tfimm runs unmodified:
``tfimm.create_model(name)(x)`` executes the reference's layer classes, block order, reshapes,
rolls, concat orders and config arithmetic, with each ``tf.*`` / ``tf.keras.layers.*`` call
evaluated in float32 by ``oracle/ops.py`` (the single place where TF semantics such as SAME
padding or the bicubic resize are restated, pinned by tests/test_oracle_ops.py).

What that pins: everything the reference's Python decides.  What it cannot pin: the numerical
behaviour of TensorFlow's own kernels beyond their published semantics.

Only ``oracle/tools/*.py`` and CPU tests import it, and only inside the build container
(/root/reference does not exist on the GPU box); its outputs travel as tests/golden fixtures.
Scope: eager inference (``training=False``) of the layers/ops the six in-scope architecture
modules use; anything else raises NotImplementedError.
"""
from . import _core
from ._core import (DType, Tensor, TensorShape, TensorSpec, Variable, as_dtype, bfloat16, cast, concat,  # noqa: F401
                    constant, convert_to_tensor, expand_dims, float16, float32, float64, floor, function, gather,
                    identity, int32, int64, name_scope, ones, ones_like, pad, rank, reduce_max, reduce_mean,
                    reduce_sum, repeat, reshape, roll, shape, split, squeeze, stack, tile, transpose, uint8, unstack,
                    where, zeros, zeros_like)
from ._core import bool_ as bool  # noqa: F401,A001
from ._core import range_ as range  # noqa: F401,A001
from . import compat, keras  # noqa: F401,E402
from .keras import initializers  # noqa: F401,E402

__version__ = "2.12.0+tfimm-oracle-standin"

import numpy as _np  # noqa: E402
import torch as _torch  # noqa: E402
from oracle import ops as _ops  # noqa: E402


def zeros_initializer():
    return keras.initializers.Zeros()


def ones_initializer():
    return keras.initializers.Ones()


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _softmax(logits, axis=-1, **_):
    return Tensor(_ops.softmax(_core._raw(logits), axis))


def _moments(x, axes, keepdims=False, **_):
    t = _core._raw(x)
    ax = tuple(int(a) for a in axes)
    m = t.mean(dim=ax, keepdim=True)
    v = ((t - m) ** 2).mean(dim=ax, keepdim=True)
    if not keepdims:
        m, v = m.squeeze(ax), v.squeeze(ax)
    return Tensor(m), Tensor(v)


def _batch_normalization(x, mean, variance, offset, scale, variance_epsilon, **_):
    r = _core._raw
    inv = _torch.rsqrt(r(variance) + variance_epsilon)
    if scale is not None:
        inv = inv * r(scale)
    y = r(x) * inv
    shift = -r(mean) * inv
    if offset is not None:
        shift = shift + r(offset)
    return Tensor(y + shift)


def _nn_depthwise_conv2d(input, filter, strides, padding, dilations=None, **_):
    assert strides[0] == 1 and strides[3] == 1
    return Tensor(_ops.depthwise_conv2d(_core._raw(input), _core._raw(filter), None, stride=(strides[1], strides[2]),
                                        padding=padding.lower()))


def _act(name):
    return getattr(keras.activations, name)


nn = _NS(softmax=_softmax, moments=_moments, batch_normalization=_batch_normalization,
         depthwise_conv2d=_nn_depthwise_conv2d, relu=_act("relu"), relu6=_act("relu6"), gelu=_act("gelu"),
         swish=_act("swish"), silu=_act("swish"), sigmoid=_act("sigmoid"), tanh=_act("tanh"))

linalg = _NS(matmul=_core.matmul)
matmul = _core.matmul

math = _NS(sqrt=_core.sqrt, rsqrt=_core.rsqrt, reduce_mean=_core.reduce_mean, reduce_sum=_core.reduce_sum,
           reduce_variance=_core.reduce_variance, reduce_max=_core.reduce_max, divide=_core.divide, floor=_core.floor,
           sigmoid=_act("sigmoid"), tanh=_act("tanh"))
sqrt = _core.sqrt


def _resize(images, size, method="bilinear", antialias=False, **_):
    if method != "bicubic" or antialias:
        raise NotImplementedError("only tf.image.resize(method='bicubic', antialias=False) is restated")
    return Tensor(_ops.resize_bicubic_tf(_core._raw(images).float(), (int(size[0]), int(size[1]))))


image = _NS(resize=_resize)

_rng = _np.random.default_rng(0)


def _uniform(shape, minval=0.0, maxval=1.0, dtype=float32, **_):
    return Tensor(_core._raw(_rng.uniform(minval, maxval, size=[int(s) for s in shape]), dtype))


def _normal(shape, mean=0.0, stddev=1.0, dtype=None, **_):
    return Tensor(_core._raw(_rng.normal(mean, stddev, size=[int(s) for s in shape]), dtype or float32))


random = _NS(uniform=_uniform, normal=_normal, set_seed=lambda s: None)


class _Err(Exception):
    pass


errors = _NS(UnknownError=type("UnknownError", (_Err,), {}), ResourceExhaustedError=type("ResourceExhaustedError", (_Err,), {}),
             InvalidArgumentError=type("InvalidArgumentError", (_Err,), {}), InternalError=type("InternalError", (_Err,), {}))


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    raise NotImplementedError(f"tf.{name} is not provided by the tfimm oracle stand-in (oracle/tf_shim)")
