"""Precision of the engine's activation / weight storage.

``bf16`` (default): the product path -- bf16 activations and GEMM weights, fp32 accumulation, the hand-written MFMA kernels.
``fp32``: the VERIFICATION path -- the same layer program (same lowering code and host-side weight transformations, minus the
cross-layer fusions) bound to the plain float32 kernels of csrc/ref32.hip.  The reference is float32 end to end and pins values
at 1e-3 relative to the maximum (tests/test_timm.py:71); this mode exists so that the engine's arithmetic can be held to that
bar (tests/test_gpu_fp32.py).  It is 20-50x slower and nothing selects it unless asked: ``TFIMM_PRECISION=fp32`` in the
environment, ``precision.set("fp32")``, or ``with precision.use("fp32"):`` around model calls.
"""
import contextlib
import os

_VALID = ("bf16", "fp32")
_current = os.environ.get("TFIMM_PRECISION", "bf16").lower()
if _current not in _VALID:
    raise ValueError(f"TFIMM_PRECISION={_current!r}: expected one of {_VALID}")


def get() -> str:
    return _current


def set(p: str) -> None:          # noqa: A001  (mirrors torch.set_default_dtype in spirit)
    global _current
    if p not in _VALID:
        raise ValueError(f"precision {p!r}: expected one of {_VALID}")
    _current = p


@contextlib.contextmanager
def use(p: str):
    prev = get()
    set(p)
    try:
        yield
    finally:
        set(prev)
