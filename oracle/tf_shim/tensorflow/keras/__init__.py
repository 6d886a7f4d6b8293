"""``tf.keras`` of the stand-in."""
from . import activations, backend, initializers, layers, models, utils  # noqa: F401
from .models import Model, Sequential  # noqa: F401


class _MixedPrecision:
    @staticmethod
    def set_global_policy(policy):
        if policy not in ("float32", None):
            raise NotImplementedError("the stand-in computes in float32 only")


mixed_precision = _MixedPrecision()
