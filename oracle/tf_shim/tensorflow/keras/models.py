"""``tf.keras.Model`` / ``tf.keras.Sequential`` (forward path only)."""
from .. import _core as C
from .layers import Layer


class Model(Layer):
    def __init__(self, *args, **kwargs):
        if args:
            raise NotImplementedError("functional-API construction is not provided by the stand-in")
        super().__init__(**kwargs)

    def predict(self, x, **_):
        out = self(x, training=False)
        return out.numpy() if hasattr(out, "numpy") else out

    def compile(self, *a, **k):
        raise NotImplementedError("the stand-in implements the inference path only")

    fit = save = save_weights = load_weights = compile

    def summary(self, print_fn=print, **_):
        for v in self.weights:
            print_fn(f"{v.name:80s} {tuple(v.shape)}")


class Sequential(Model):
    def __init__(self, layers=None, name=None):
        super().__init__(name=name)
        self._seq = []
        for l in layers or []:
            self.add(l)

    def add(self, layer):
        self._seq.append(layer)

    @property
    def layers(self):
        return list(self._seq)

    def call(self, inputs, training=None, mask=None):
        x = inputs
        # Keras builds the layers of a deferred-build Sequential in a scratch functional graph: the name
        # scopes around the Sequential (and its own) are not part of their variables' names.
        with C.fresh_name_scope():
            for l in self._seq:
                if l._call_has_training:
                    x = l(x, training=training)
                else:
                    x = l(x)
        return x


def load_model(*a, **k):
    raise NotImplementedError("SavedModel loading is not provided by the stand-in (no TensorFlow)")


def model_from_config(*a, **k):
    raise NotImplementedError("model_from_config is not provided by the stand-in")
