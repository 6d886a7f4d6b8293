"""Position-embedding interpolation (reference tfimm/layers/transformers.py:13-47).

The reference calls ``tf.image.resize(method="bicubic")`` on the (1, gh, gw, D) grid of embeddings.
TensorFlow is not available here, so the op is restated from its published kernel
(tensorflow/core/kernels/image/resize_bicubic_op.cc, half-pixel centres, ``antialias=False``):

  * source coordinate of output pixel o:  (o + 0.5) * in/out - 0.5;
  * Keys cubic convolution kernel with A = -0.5, evaluated from a 1024-entry table: the fractional
    part is rounded to a multiple of 1/1024 (``lrintf(delta * 1024)``);
  * taps that fall outside the image get weight 0 and the remaining weights are renormalised
    (no edge replication);
  * rows and columns are separable.

This runs once per (model, input size) on a few hundred vectors, on the host, in float32.
"""
from typing import Tuple

import numpy as np

_TABLE = 1024
_A = np.float32(-0.5)


def _coeff_table():
    x = (np.arange(_TABLE + 1, dtype=np.float32) / np.float32(_TABLE)).astype(np.float32)
    near = ((_A + 2) * x - (_A + 3)) * x * x + 1            # |d| <= 1
    xf = x + 1
    far = ((_A * xf - 5 * _A) * xf + 8 * _A) * xf - 4 * _A   # 1 < |d| < 2
    return near.astype(np.float32), far.astype(np.float32)


def _axis_weights(n_in: int, n_out: int):
    """(n_out, 4) tap indices (clamped) and weights of one axis."""
    near, far = _coeff_table()
    scale = np.float32(n_in) / np.float32(n_out)
    idx = np.zeros((n_out, 4), np.int64)
    wgt = np.zeros((n_out, 4), np.float32)
    for o in range(n_out):
        loc = (np.float32(o) + np.float32(0.5)) * scale - np.float32(0.5)
        base = int(np.floor(loc))
        off = int(np.rint((loc - np.float32(base)) * _TABLE))
        taps = (base - 1, base, base + 1, base + 2)
        raw = (far[off], near[off], near[_TABLE - off], far[_TABLE - off])
        for t, (i, w) in enumerate(zip(taps, raw)):
            inside = 0 <= i < n_in
            idx[o, t] = min(max(i, 0), n_in - 1)
            wgt[o, t] = w if inside else 0.0
        s = wgt[o].sum()
        if abs(s) >= 1000.0 * np.finfo(np.float32).tiny:
            wgt[o] /= s
    return idx, wgt


def resize_bicubic(x: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """``tf.image.resize(x, size, method="bicubic")`` for x of shape (B, H, W, C), float32."""
    x = np.asarray(x, dtype=np.float32)
    _, H, W, _ = x.shape
    iy, wy = _axis_weights(H, int(size[0]))
    ix, wx = _axis_weights(W, int(size[1]))
    rows = np.einsum("ot,botwc->bowc", wy, x[:, iy])            # (B, OH, W, C)
    return np.einsum("pt,bhptc->bhpc", wx, rows[:, :, ix]).astype(np.float32)


def interpolate_pos_embeddings(pos_embed, src_grid_size: Tuple[int, int], tgt_grid_size: Tuple[int, int],
                               nb_tokens: int = 0) -> np.ndarray:
    """Resize (1, nb_tokens + gh*gw, D) position embeddings to another patch grid; the first
    ``nb_tokens`` rows (class / distillation tokens) are carried over unchanged
    (reference layers/transformers.py:13-47)."""
    pos_embed = np.asarray(pos_embed, dtype=np.float32)
    if tuple(src_grid_size) == tuple(tgt_grid_size):
        return pos_embed
    grid = pos_embed[:, nb_tokens:].reshape(1, src_grid_size[0], src_grid_size[1], -1)
    out = resize_bicubic(grid, tgt_grid_size).reshape(1, tgt_grid_size[0] * tgt_grid_size[1], -1)
    return np.concatenate((pos_embed[:, :nb_tokens], out), axis=1)
