"""``tf.keras.backend`` / ``tensorflow.python.keras.backend`` (models/factory.py:7,285; utils/timm.py:24)."""
from . import layers as _layers


def floatx():
    return "float32"


def batch_set_value(tuples):
    for var, value in tuples:
        var.assign(value.numpy() if hasattr(value, "numpy") else value)


def batch_get_value(tensors):
    return [t.numpy() for t in tensors]


def clear_session():
    _layers.reset_uids()
