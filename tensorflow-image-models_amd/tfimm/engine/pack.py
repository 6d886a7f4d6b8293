"""Host-side weight packing: tfimm-named fp32 arrays -> the layouts the HIP kernels read.

Everything here is numpy on the host (testable without a GPU); ``graph.Program.upload``
moves the results to HBM.

Layouts (see include/tfimm_hip.h):
  * GEMM / conv weight  ``Wt[N][ldw]`` bf16, K contiguous, zero padded to ldw = ceil8(K).
    conv K order is (ky, kx, ci) = C-order flatten of Keras' HWIO kernel
    (reference weight layout: tfimm/utils/timm.py:164-170).
  * inference BatchNorm is folded:  w' = w * s,  b' = beta - mean * s,
    s = gamma / sqrt(var + eps)   (every conv in front of a BN has
    use_bias=False: resnet.py:223,236,246, efficientnet_blocks.py:32).
"""
from typing import Optional, Tuple

import numpy as np


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit pattern (uint16), round-to-nearest-even (same as the kernels' f2bf)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return rounded.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def bn_scale_shift(gamma, beta, mean, var, eps) -> Tuple[np.ndarray, np.ndarray]:
    s = gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps)
    return s.astype(np.float32), (beta.astype(np.float64) - mean.astype(np.float64) * s).astype(np.float32)


def pad_channels(cin: int) -> int:
    """Channel count the input image is stored with on the device."""
    return 4 if cin <= 4 else ceil_to(cin, 8)


def pack_matrix(w_nk: np.ndarray) -> np.ndarray:
    """[N][K] fp32 -> [N][ceil64(K)] bf16 bits, zero padded."""
    n, k = w_nk.shape
    out = np.zeros((n, ceil_to(k, 64)), dtype=np.uint16)  # 64: one whole LDS-DMA K-tile
    out[:, :k] = to_bf16_bits(w_nk)
    return out


def pack_dense(kernel: np.ndarray, bias: Optional[np.ndarray]):
    """Keras Dense kernel (in, out) -> Wt[out][in]."""
    wt = pack_matrix(np.ascontiguousarray(kernel.T))
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    return wt, b


def pack_conv(kernel: np.ndarray, scale: Optional[np.ndarray], shift: Optional[np.ndarray],
              cin_stored: int):
    """HWIO conv kernel -> (Wt, bias, K, mode).

    ``cin_stored`` is the channel count of the activation tensor in HBM (4 for padded RGB).
    mode 1 = generic gather, 2 = Cin==4 pixel-pair gather (kx padded to even).
    """
    kh, kw, cin, cout = kernel.shape
    k = kernel.astype(np.float32)
    if scale is not None:
        k = k * scale.reshape(1, 1, 1, cout)
    if cin_stored == 4:
        assert cin <= 4
        kwp = (kw + 1) // 2 * 2
        kp = np.zeros((kh, kwp, 4, cout), dtype=np.float32)
        kp[:, :kw, :cin, :] = k
        mode, kk = 2, kh * kwp * 4
    else:
        if cin_stored != cin:
            kp = np.zeros((kh, kw, cin_stored, cout), dtype=np.float32)
            kp[:, :, :cin, :] = k
        else:
            kp = k
        mode, kk = 1, kh * kw * cin_stored
    wt = pack_matrix(np.ascontiguousarray(kp.reshape(kk, cout).T))
    b = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    return wt, b, kk, mode


def pack_depthwise(kernel: np.ndarray, scale: Optional[np.ndarray], shift: Optional[np.ndarray]):
    """Keras depthwise kernel (kh, kw, C, 1) -> fp32 [kh*kw][C] (+ folded scale), bias."""
    kh, kw, c, mult = kernel.shape
    assert mult == 1
    w = kernel[..., 0].astype(np.float32).reshape(kh * kw, c)
    if scale is not None:
        w = w * scale.reshape(1, c)
    b = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    return np.ascontiguousarray(w), b
