from tensorflow.keras import backend  # noqa: F401
