"""``tf.keras.utils``: serialisation hooks are no-ops (models/serialization.py:86-88)."""
_registered = {}


def register_keras_serializable(package="Custom", name=None):
    def deco(cls):
        _registered[f"{package}>{name or cls.__name__}"] = cls
        return cls
    return deco


def serialize_keras_object(obj):
    return {"class_name": type(obj).__name__, "config": obj.get_config() if hasattr(obj, "get_config") else {}}


def get_custom_objects():
    return _registered
