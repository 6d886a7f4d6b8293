"""fp32 CPU restatement of reference tfimm/architectures/swin.py (test infrastructure).

Follows SwinTransformer.forward_features/call (swin.py:488-517), SwinTransformerStage.call
(:399-407), SwinTransformerBlock.build/call (:243-327), WindowAttention.build/call (:124-198),
PatchMerging.call (:348-362), window_partition / window_reverse (:72-108) literally: roll,
partition, gather of the bias table, the -100 mask built from slices, reverse, roll back.

Parity note: the reference's own timm comparison is disabled for Swin (tests/test_timm.py:29-30),
so this restatement is pinned by the reference's model code itself, run over the stand-in TensorFlow of
oracle/tf_shim (oracle/tools/make_reference_golden.py -> tests/golden/forward_golden.npz: swin_test_model,
swin_shift_test_model incl. every feature, swin_tiny / swin_base logits; tests/test_golden.py).
"""
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .common import LN_EPS, W, finish, mlp


def window_partition(x, ws):                                            # swin.py:72-87
    b, h, w, c = x.shape
    x = x.reshape(-1, h // ws, ws, w // ws, ws, c)
    x = x.permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, c)


def window_reverse(windows, ws, h, w, c):                               # swin.py:90-108
    x = windows.reshape(-1, h // ws, w // ws, ws, ws, c)
    x = x.permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, h, w, c)


def _attn_mask(h, w, ws, shift):                                        # swin.py:248-273
    img_mask = np.zeros([1, h, w, 1])
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img_mask[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(torch.from_numpy(img_mask), ws).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    m = torch.where(m != 0, torch.full_like(m, -100.0), m)
    return m.float()                                                     # (nW, n, n)


def _rel_index(ws):                                                     # swin.py:143-152
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij"))
    flat = coords.reshape(2, -1)
    rel = (flat[:, :, None] - flat[:, None, :]).transpose((1, 2, 0)).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).astype(np.int64)


def _window_attention(w: W, cfg, x, mask, prefix, embed_dim, nb_heads):  # swin.py:159-198
    ws = cfg.window_size
    _, n, c = x.shape
    qkv = w.dense(x, prefix + "/qkv", bias=cfg.qkv_bias)
    qkv = qkv.reshape(-1, n, 3, nb_heads, c // nb_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    scale = (embed_dim // nb_heads) ** -0.5
    q = q * scale                                                        # :172-173 scale BEFORE the matmul
    attn = q @ k.transpose(-1, -2)
    table = w(prefix + "/relative_position_bias_table")
    bias = table[torch.from_numpy(_rel_index(ws).reshape(-1))]           # tf.gather :175-178
    bias = bias.reshape(ws ** 2, ws ** 2, -1).permute(2, 0, 1)           # :179-183
    attn = attn + bias.unsqueeze(0)
    nw = mask.shape[0]
    if mask.dim() == 1:                                                  # zeros((1,)) broadcast :283-285
        attn = attn + mask.reshape(1, 1, 1, 1)
    else:
        attn = attn.reshape(-1, nw, nb_heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.reshape(-1, nb_heads, n, n)
    attn = ops.softmax(attn, -1)
    x = (attn @ v).permute(0, 2, 1, 3).reshape(-1, n, c)
    return w.dense(x, prefix + "/proj")


def swin_forward(cfg, weights, x, return_features=False):
    w = W(weights)
    eps = LN_EPS[cfg.norm_layer]
    x = ops.as_t(x)
    feats = OrderedDict()
    B = x.shape[0]
    # PatchEmbeddings (norm_layer = cfg.norm_layer)
    x = ops.conv2d(x, w("patch_embed/proj/kernel"), w("patch_embed/proj/bias"), stride=cfg.patch_size)
    x = x.reshape(B, -1, x.shape[-1])
    x = w.ln(x, "patch_embed/norm", eps)
    feats["patch_embedding"] = x
    nst = len(cfg.nb_blocks)
    k = 0
    for i in range(nst):
        h, wd = cfg.patch_resolution[0] // 2 ** i, cfg.patch_resolution[1] // 2 ** i
        D, nh = int(cfg.embed_dim * 2 ** i), cfg.nb_heads[i]
        for j in range(cfg.nb_blocks[i]):
            p = f"layers/{i}/blocks/{j}"
            shift = 0 if j % 2 == 0 else cfg.window_size // 2           # :387
            ws = cfg.window_size
            if min(h, wd) <= ws:                                         # :221-223
                shift, ws = 0, min(h, wd)
            mask = _attn_mask(h, wd, ws, shift) if shift > 0 else torch.zeros(1)
            c = x.shape[-1]
            shortcut = x
            y = w.ln(x, p + "/norm1", eps).reshape(-1, h, wd, c)         # :295-296
            y = ops.roll(y, (-shift, -shift), (1, 2))                    # :299
            y = window_partition(y, ws).reshape(-1, ws * ws, c)          # :302-303
            y = _window_attention(w, cfg, y, mask, p + "/attn", D, nh)   # :306
            y = y.reshape(-1, ws, ws, c)
            y = window_reverse(y, ws, h, wd, c)                          # :309-310
            y = ops.roll(y, (shift, shift), (1, 2)).reshape(-1, h * wd, c)   # :313-314
            x = y + shortcut                                             # :317-318
            shortcut = x
            y = mlp(w, w.ln(x, p + "/norm2", eps), p + "/mlp", cfg.act_layer)
            x = y + shortcut                                             # :322-325
            feats[f"block_{k}"] = x
            k += 1
        if i < nst - 1:                                                  # PatchMerging.call :348-362
            c = x.shape[-1]
            t = x.reshape(-1, h, wd, c)
            t = torch.cat((t[:, 0::2, 0::2, :], t[:, 1::2, 0::2, :], t[:, 0::2, 1::2, :], t[:, 1::2, 1::2, :]), dim=-1)
            t = t.reshape(-1, (h // 2) * (wd // 2), 4 * c)
            t = w.ln(t, f"layers/{i}/downsample/norm", eps)
            x = ops.dense(t, w(f"layers/{i}/downsample/reduction/kernel"))
        feats[f"stage_{i}"] = x
    x = w.ln(x, "norm", eps)
    feats["features_all"] = x
    x = ops.global_avg_pool(x)                                           # GlobalAveragePooling1D
    feats["features"] = x
    if cfg.nb_classes > 0:
        x = w.dense(x, "head")
    feats["logits"] = x
    return finish(x, feats, return_features)
