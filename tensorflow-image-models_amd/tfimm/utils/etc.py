"""Small arithmetic helpers that decide layer shapes.

Behavioural mirror of reference tfimm/utils/etc.py:7-26 (``to_2tuple``,
``make_divisible``) and tfimm/layers/conv.py:15-28 (``get_padding``).
"""
from collections.abc import Iterable


def to_2tuple(x):
    if isinstance(x, Iterable):
        return tuple(x)[:2]
    return (x, x)


def make_divisible(value, divisor, min_value=None, round_limit=0.9):
    """Round ``value`` to a multiple of ``divisor`` without losing more than
    ``1 - round_limit`` of it (drives EfficientNet channel counts)."""
    floor = min_value or divisor
    rounded = max(floor, int(value + divisor / 2) // divisor * divisor)
    if rounded < round_limit * value:
        rounded += divisor
    return rounded


def get_padding(kernel_size, strides=1, dilation_rate=1):
    """PyTorch-style symmetric padding, per spatial axis."""
    k, s, d = to_2tuple(kernel_size), to_2tuple(strides), to_2tuple(dilation_rate)
    return tuple(((s[i] - 1) + d[i] * (k[i] - 1)) // 2 for i in range(2))


def same_padding(in_size, kernel, stride, dilation=1):
    """TF ``padding="same"`` for one axis -> (out_size, pad_before, pad_after).

    out = ceil(in / s); total = max((out-1)*s + k_eff - in, 0); the extra pixel goes
    after (bottom/right) -- SURVEY.md App. A.
    """
    k_eff = (kernel - 1) * dilation + 1
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    before = total // 2
    return out, before, total - before
