"""``tf.keras.activations`` by name (layers/factory.py:6-13 uses linear/swish/relu/gelu/sigmoid;
vit.py:357-359 uses tanh).  Keras' ``gelu`` defaults to approximate=False, i.e. the exact erf form."""
from oracle import ops as _ops

from .._core import Tensor, _raw


def _mk(name):
    def fn(x):
        return Tensor(_ops.activation(_raw(x), name))
    fn.__name__ = name
    return fn


linear = _mk("linear")
relu = _mk("relu")
relu6 = _mk("relu6")
gelu = _mk("gelu")
swish = _mk("swish")
silu = swish
sigmoid = _mk("sigmoid")
tanh = _mk("tanh")


def softmax(x, axis=-1):
    return Tensor(_ops.softmax(_raw(x), axis))


_BY_NAME = {"linear": linear, "relu": relu, "relu6": relu6, "gelu": gelu, "swish": swish, "silu": swish,
            "sigmoid": sigmoid, "tanh": tanh, "softmax": softmax}


def get(identifier):
    if identifier is None:
        return linear
    if callable(identifier):
        return identifier
    if identifier not in _BY_NAME:
        raise ValueError(f"Unknown activation function: {identifier}")
    return _BY_NAME[identifier]
