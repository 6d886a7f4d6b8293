"""Dump every model config the reference registers, as JSON (test fixture generator).

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference).  The
reference imports TensorFlow at module scope, and TensorFlow is not installable here, so a
*permissive stand-in* `tensorflow` module is injected: class statements and decorators work,
nothing is ever executed.  ``@register_model`` functions only construct dataclasses, so the
registry the reference would build is reproduced exactly.

Output: tests/golden/reference_configs.json  {model name: {"cls": ..., "module": ..., "cfg": {...}}}
"""
import dataclasses
import json
import os
import sys
import types


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make(name)

    def __call__(cls, *a, **k):
        # decorator use: @tf.keras.utils.register_keras_serializable(...)(cls) etc.
        if len(a) == 1 and not k and isinstance(a[0], type):
            return a[0]
        return type.__call__(cls, *a, **k)


def _make(name):
    return _Meta(name, (Stub,), {})


class Stub(metaclass=_Meta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and (isinstance(a[0], type) or callable(a[0])):
            return a[0]
        return Stub()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Stub()

    def __mro_entries__(self, bases):
        return (Stub,)

    def __iter__(self):
        return iter(())


class _Module(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        return _make(name)


def install_stub_tensorflow():
    for n in ["tensorflow", "tensorflow.keras", "tensorflow.python", "tensorflow.python.keras",
              "tensorflow.python.keras.backend", "tensorflow.python.framework",
              "tensorflow.python.framework.convert_to_constants", "tensorflow.keras.layers",
              "tensorflow.keras.backend", "tensorflow.compat", "tensorflow.compat.v1"]:
        sys.modules[n] = _Module(n)
    sys.modules["tensorflow"].__version__ = "0.0-stub"


def main(out_path):
    install_stub_tensorflow()
    sys.path.insert(0, "/root/reference")
    import tfimm  # noqa: F401  (registers everything)
    from tfimm.models import registry

    out = {}
    for name in sorted(registry._model_class):
        cfg = registry._model_config[name]
        d = dataclasses.asdict(cfg)
        module = [m for m, names in registry._module_to_models.items() if name in names][0]
        out[name] = {"cls": registry._model_class[name].__name__, "module": module, "cfg": d}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True, default=lambda o: list(o) if isinstance(o, tuple) else str(o))
    print(f"{len(out)} configs -> {out_path}")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(os.path.join(here, "..", "..", "tests", "golden", "reference_configs.json"))
