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


def pack_matrix(w_nk: np.ndarray, fp32: bool = False) -> np.ndarray:
    """[N][K] fp32 -> [N][ceil64(K)] bf16 bits, zero padded (``fp32``: the same layout in float32 -- the verification
    path of engine/precision.py keeps the weights unrounded)."""
    n, k = w_nk.shape
    if fp32:
        out32 = np.zeros((n, ceil_to(k, 64)), dtype=np.float32)
        out32[:, :k] = w_nk
        return out32
    out = np.zeros((n, ceil_to(k, 64)), dtype=np.uint16)  # 64: one whole LDS-DMA K-tile
    out[:, :k] = to_bf16_bits(w_nk)
    return out


def pack_dense(kernel: np.ndarray, bias: Optional[np.ndarray], fp32: bool = False):
    """Keras Dense kernel (in, out) -> Wt[out][in]."""
    wt = pack_matrix(np.ascontiguousarray(kernel.T), fp32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    return wt, b


def pack_conv(kernel: np.ndarray, scale: Optional[np.ndarray], shift: Optional[np.ndarray],
              cin_stored: int, fp32: bool = False):
    """HWIO conv kernel -> (Wt, bias, K, mode).

    ``cin_stored`` is the channel count of the activation tensor in HBM (4 for padded RGB).
    mode 1 = generic gather, 2 = Cin==4 pixel-pair gather (kx padded to even).
    """
    kh, kw, cin, cout = kernel.shape
    k = kernel.astype(np.float32)
    if scale is not None:
        k = k * scale.reshape(1, 1, 1, cout)
    if cin_stored == 4 and not fp32:
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
    wt = pack_matrix(np.ascontiguousarray(kp.reshape(kk, cout).T), fp32)
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


def swin_bias_tiles(rel_bias: np.ndarray, window: int, shift: int) -> np.ndarray:
    """(rel_bias + shift mask) * log2(e) per window kind -> fp32 [kinds][heads][n][ceil64(n)].

    ``rel_bias``: gathered relative_position_bias [heads][n][n] (swin.py:175-184).  The mask of
    swin.py:249-273 gives a token of the shifted frame the region id 3*rh + rw with
    rh = 0 / 1 / 2 for rows in [0, H-ws) / [H-ws, H-shift) / [H-shift, H) (rw alike), and adds -100
    between tokens of different regions.  Inside one window only the LAST window row / column sees
    more than one region, so there are four distinct patterns: kind = 2*last_row + last_col.
    """
    heads, n, _ = rel_bias.shape
    nkp = ceil_to(n, 64)
    kinds = 4 if shift > 0 else 1
    log2e = 1.4426950408889634
    out = np.zeros((kinds, heads, n, nkp), dtype=np.float32)
    ty, tx = np.divmod(np.arange(n), window)
    for kind in range(kinds):
        last_row, last_col = kind >> 1, kind & 1
        rh = np.where(ty < window - shift, 1, 2) if last_row else np.zeros(n, dtype=np.int64)
        rw = np.where(tx < window - shift, 1, 2) if last_col else np.zeros(n, dtype=np.int64)
        reg = rh * 3 + rw
        mask = np.where(reg[:, None] != reg[None, :], -100.0, 0.0) if shift > 0 else 0.0
        out[kind, :, :, :n] = (rel_bias.astype(np.float64) + mask) * log2e
    return out


def chain_k_order(c1: int) -> np.ndarray:
    """K-axis order of the second GEMM of tfimm_hip_conv_chain: position 16 t + s holds channel
    16 t + (0..3, 8..11, 4..7, 12..15)[s] -- the order in which a wave's GEMM-1 accumulators (one pixel per lane, channel
    quads 8 q + 4 (lane >> 5)) line up as the 8-consecutive-k register operand of v_mfma_f32_32x32x16_bf16."""
    assert c1 % 16 == 0
    inner = np.array([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    return (np.arange(0, c1, 16)[:, None] + inner[None, :]).reshape(-1)


def pack_mlp_fused(k1: np.ndarray, b1: Optional[np.ndarray], gamma: np.ndarray, beta: np.ndarray, k2: np.ndarray,
                   b2: Optional[np.ndarray], out_scale: Optional[np.ndarray] = None):
    """Operands of tfimm_hip_mlp_fused for LayerNorm(gamma, beta) -> Dense k1 (C, H) + b1 -> act -> Dense k2 (H, C) + b2
    (-> * out_scale): (w1 uint16 [H][C] with gamma folded, b1' f32 [H] = beta . k1 + b1, w2 uint16 [C][H] with its K axis
    in chain_k_order and out_scale folded, b2' f32 [C])."""
    c, h = k1.shape
    assert k2.shape == (h, c) and gamma.shape == (c,) and beta.shape == (c,)
    k1d = k1.astype(np.float64)
    shift = beta.astype(np.float64) @ k1d
    b1f = (shift if b1 is None else shift + b1.astype(np.float64)).astype(np.float32)
    w1, _ = pack_dense((k1d * gamma.astype(np.float64).reshape(c, 1)).astype(np.float32), None)
    k2d = k2.astype(np.float64)
    b2d = np.zeros(c, np.float64) if b2 is None else b2.astype(np.float64)
    if out_scale is not None:
        g = out_scale.astype(np.float64).reshape(c)
        k2d, b2d = k2d * g.reshape(1, c), b2d * g
    w2, _ = pack_dense(k2d[chain_k_order(h)].astype(np.float32), None)
    return (np.ascontiguousarray(w1[:h, :c]), b1f, np.ascontiguousarray(w2[:c, :h]), b2d.astype(np.float32))


def pack_grouped3x3(kernel: np.ndarray, groups: int, scale: Optional[np.ndarray]) -> np.ndarray:
    """Grouped 3x3 kernel (3, 3, C / groups, C), input and output width of a group equal and <= 32 -> the per-lane MFMA A
    fragments tfimm_hip_grouped_conv3x3 keeps in registers: uint16 [C / 32][18][64][8].  Super-group sg = output channels
    [32 sg, 32 sg + 32); its inputs are the same 32 channels.  Fragment ks = 2 tap + half, lane l: output channel
    32 sg + (l & 31), input channels 32 sg + 16 half + 8 (l >> 5) + 0..7 (zero where the two are in different groups)."""
    kh, kw, w, c = kernel.shape
    assert (kh, kw) == (3, 3) and c == w * groups and w <= 32 and 32 % w == 0 and c % 32 == 0, kernel.shape
    k = kernel.astype(np.float32)
    if scale is not None:
        k = k * scale.reshape(1, 1, 1, c)
    nsg = c // 32
    dense = np.zeros((nsg, 9, 32, 32), dtype=np.float32)          # [sg][tap][out i][in cl]
    for sg in range(nsg):
        for i in range(32):
            n = sg * 32 + i
            g = n // w
            lo = g * w - sg * 32                                      # first local input channel of that group
            dense[sg, :, i, lo:lo + w] = k[:, :, :, n].reshape(9, w)
    lane = np.arange(64)
    out = np.zeros((nsg, 18, 64, 8), dtype=np.float32)
    for tap in range(9):
        for half in range(2):
            cl = half * 16 + (lane >> 5)[:, None] * 8 + np.arange(8)[None, :]      # (64, 8)
            out[:, tap * 2 + half] = dense[:, tap][:, (lane & 31)[:, None], cl]
    return to_bf16_bits(out).reshape(nsg, 18, 64, 8)


def pack_expand_frag(w1: np.ndarray, cpad: int) -> np.ndarray:
    """Expand (1x1) weights ``w1[Cin][C]`` (BN scale folded, Cin <= 32) as the MFMA A fragments
    tfimm_hip_expand_dwconv loads: uint16 [cpad/32][2][64][8], element [cc][ks][lane][j] =
    w1[16 ks + 8 (lane >> 5) + j][32 cc + (lane & 31)], zero beyond Cin / C."""
    cin, c = w1.shape
    assert cin <= 32 and cpad % 32 == 0 and cpad >= c
    full = np.zeros((32, cpad), np.float32)
    full[:cin, :c] = w1
    lane = np.arange(64)
    k = (16 * np.arange(2)[:, None, None] + 8 * (lane >> 5)[None, :, None] + np.arange(8)[None, None, :])    # [2][64][8]
    ch = 32 * np.arange(cpad // 32)[:, None, None, None] + (lane & 31)[None, None, :, None]                # [cc][1][64][1]
    return to_bf16_bits(full[k[None], ch])


def split3_bf16(v: np.ndarray) -> np.ndarray:
    """fp32 values as three bf16 terms (truncation splits with exact residuals): v ~= t0 + t1 + t2 to ~24 bits.
    Returns uint16 [..., 3]."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.zeros(v.shape + (3,), np.uint16)
    r = v.copy()
    for i in range(3):
        bits = r.view(np.uint32) & np.uint32(0xFFFF0000)
        out[..., i] = (bits >> 16).astype(np.uint16)
        r = r - bits.view(np.float32)
    return out


def pack_ln_c1(wt_bits: np.ndarray, n: int, k: int) -> np.ndarray:
    """Correction fragments of a GEMM with a folded LayerNorm (tfimm_gemm_desc.ln_c1): c1[n] = sum_k Wt[n][k] over the
    bf16-ROUNDED packed weights ``wt_bits`` (uint16 [N][ldw]), split into three bf16 terms (ca, cb, cc) and laid out per column
    as the MFMA fragment pair {ca, cb, cc, ca, cb, cc, ca, cb} {cc, 0, 0, 0, 0, 0, 0, 0}: uint16 [N][2][8]."""
    c1 = bf16_bits_to_f32(wt_bits[:n, :k]).astype(np.float64).sum(axis=1).astype(np.float32)
    t = split3_bf16(c1)                                     # [N][3]
    out = np.zeros((n, 2, 8), np.uint16)
    out[:, 0, :] = t[:, [0, 1, 2, 0, 1, 2, 0, 1]]
    out[:, 1, 0] = t[:, 2]
    return out


def pack_stem_frag(kernel: np.ndarray, cpad: int) -> np.ndarray:
    """3 x 3 stem kernel (3, 3, cin <= 4, C) (BN scale folded) as the MFMA A fragments of tfimm_hip_expand_dwconv's stem
    flavour: uint16 [cpad/32][3][64][8], element [cc][ks][lane][j] = kernel[tap // 3][tap % 3][j % 4][32 cc + (lane & 31)]
    with tap = 4 ks + 2 (lane >> 5) + j // 4 (zero for tap >= 9, channels the kernel does not have, columns >= C)."""
    kh, kw, cin, c = kernel.shape
    assert (kh, kw) == (3, 3) and cin <= 4 and cpad % 32 == 0 and cpad >= c
    full = np.zeros((12, 4, cpad), np.float32)                     # [tap][channel][column]
    full[:9, :cin, :c] = kernel.reshape(9, cin, c)
    lane = np.arange(64)
    j = np.arange(8)
    tap = 4 * np.arange(3)[:, None, None] + 2 * (lane >> 5)[None, :, None] + (j // 4)[None, None, :]       # [3][64][8]
    ch4 = np.broadcast_to((j % 4)[None, None, :], tap.shape)
    col = 32 * np.arange(cpad // 32)[:, None, None, None] + (lane & 31)[None, None, :, None]                # [cc][1][64][1]
    return to_bf16_bits(full[tap[None], ch4[None], col])


def pack_chain_ds(w: np.ndarray) -> np.ndarray:
    """Shortcut 1x1 convolution of a bottleneck block, ``w[64][N2]`` (BN scale folded), as the MFMA A fragments the fused
    tail kernel multiplies with the block input (tfimm_chain_desc.ds_w): uint16 [N2/32][4][64][8], element
    [blk][t][lane][e] = w[16 t + 8 (lane >> 5) + e][32 blk + (lane & 31)]."""
    cin, n2 = w.shape
    assert cin == 64 and n2 % 32 == 0
    lane = np.arange(64)
    k = 16 * np.arange(4)[:, None, None] + 8 * (lane >> 5)[None, :, None] + np.arange(8)[None, None, :]      # [4][64][8]
    col = 32 * np.arange(n2 // 32)[:, None, None, None] + (lane & 31)[None, None, :, None]                   # [blk][1][64][1]
    return to_bf16_bits(np.asarray(w, np.float32)[k[None], col])
