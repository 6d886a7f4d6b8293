"""Host-side helpers that the reference exposes from ``tfimm.layers`` and that callers of the forward
path use directly (reference tfimm/layers/__init__.py).  The layers themselves are kernels of the
engine; what remains a plain function is the position-embedding resize."""
from .transformers import interpolate_pos_embeddings, resize_bicubic  # noqa: F401
