"""fp32 CPU restatement of reference tfimm/architectures/cait.py (test infrastructure).

Follows CaiT.forward_features / CaiT.call (cait.py:402-445), LayerScaleBlock.call (:311-326),
TalkingHeadAttention.call (:233-262), LayerScaleBlockClassAttention.call (:186-202),
ClassAttention.call (:118-146), PatchEmbeddings.call (layers/transformers.py:142-173) and
MLP.call (:208-214) statement by statement.
"""
from collections import OrderedDict

import torch

from . import ops
from .common import LN_EPS, W, finish, mlp


def _talking_head_attention(w: W, x, prefix, nb_heads, qkv_bias):
    # cait.py:233-262
    B, N, D = x.shape
    scale = (D // nb_heads) ** -0.5                                   # :216
    qkv = w.dense(x, prefix + "/qkv", bias=qkv_bias)                  # :237  (B, N, 3D)
    qkv = qkv.reshape(B, N, 3, nb_heads, -1).permute(2, 0, 3, 1, 4)   # :238-239 (3, B, H, N, hd)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = scale * q                                                     # :241 scale BEFORE the matmul
    attn = q @ k.transpose(-1, -2)                                    # :243 (B, H, N, N)
    attn = attn.permute(0, 2, 3, 1)                                   # :244 (B, N, N, H)
    attn = w.dense(attn, prefix + "/proj_l")                          # :245 Dense over the head axis
    attn = attn.permute(0, 3, 1, 2)                                   # :246
    attn = ops.softmax(attn, -1)                                      # :247
    attn = attn.permute(0, 2, 3, 1)                                   # :248
    attn = w.dense(attn, prefix + "/proj_w")                          # :249
    attn = attn.permute(0, 3, 1, 2)                                   # :250
    y = attn @ v                                                      # :253 (B, H, N, hd)
    y = y.permute(0, 2, 1, 3).reshape(B, N, -1)                       # :254-255
    return w.dense(y, prefix + "/proj")                               # :257


def _class_attention(w: W, x, prefix, nb_heads, qkv_bias):
    # cait.py:118-146
    B, N, D = x.shape
    scale = (D // nb_heads) ** -0.5                                   # :106
    q = w.dense(x[:, 0], prefix + "/q", bias=qkv_bias)                # :122 (B, D)
    q = q.reshape(B, 1, nb_heads, -1).permute(0, 2, 1, 3) * scale     # :123-126 (B, H, 1, hd)
    k = w.dense(x, prefix + "/k", bias=qkv_bias)                      # :128
    k = k.reshape(B, N, nb_heads, -1).permute(0, 2, 1, 3)             # :129-130
    v = w.dense(x, prefix + "/v", bias=qkv_bias)                      # :132
    v = v.reshape(B, N, nb_heads, -1).permute(0, 2, 1, 3)             # :133-134
    attn = q @ k.transpose(-1, -2)                                    # :136 (B, H, 1, N)
    attn = ops.softmax(attn, -1)                                      # :137
    y = attn @ v                                                      # :140 (B, H, 1, hd)
    y = y.permute(0, 2, 1, 3).reshape(B, 1, -1)                       # :141-142
    return w.dense(y, prefix + "/proj")                               # :144


def cait_forward(cfg, weights, x, return_features=False):
    w = W(weights)
    eps = LN_EPS[cfg.norm_layer]
    x = ops.as_t(x)
    feats = OrderedDict()
    B = x.shape[0]
    x = ops.conv2d(x, w("patch_embed/proj/kernel"), w("patch_embed/proj/bias"), stride=cfg.patch_size)
    grid = (x.shape[1], x.shape[2])
    x = x.reshape(B, -1, x.shape[-1])                                 # transformers.py:167-170
    pos = w("pos_embed")
    if cfg.interpolate_input:                                         # cait.py:407-415
        pos = ops.interpolate_pos_embeddings(pos, cfg.grid_size, grid, 0)
    x = x + pos                                                       # cait.py:406
    feats["patch_embedding"] = x
    for j in range(cfg.nb_blocks):                                    # :417-419, LayerScaleBlock.call :311-326
        p = f"blocks/{j}"
        shortcut = x
        y = w.ln(x, p + "/norm1", eps)
        y = _talking_head_attention(w, y, p + "/attn", cfg.nb_heads, cfg.qkv_bias)
        x = w(p + "/gamma_1") * y + shortcut
        shortcut = x
        y = w.ln(x, p + "/norm2", eps)
        y = mlp(w, y, p + "/mlp", cfg.act_layer)
        x = w(p + "/gamma_2") * y + shortcut
        feats[f"block_{j}"] = x
    cls = w("cls_token").expand(B, -1, -1)                            # :422 tf.repeat
    x = torch.cat((cls, x), dim=1)                                    # :423
    feats["features_cls_token"] = x
    for j in range(2):                                                # :426-428, :186-202
        p = f"blocks_token_only/{j}"
        x_cls = x[:, :1]
        u = w.ln(x, p + "/norm1", eps)
        u = w(p + "/gamma_1") * _class_attention(w, u, p + "/attn", cfg.nb_heads, cfg.qkv_bias)
        x_cls = x_cls + u
        shortcut = x_cls
        x_cls = w.ln(x_cls, p + "/norm2", eps)
        x_cls = mlp(w, x_cls, p + "/mlp", cfg.act_layer)
        x_cls = w(p + "/gamma_2") * x_cls + shortcut
        x = torch.cat((x_cls, x[:, 1:]), dim=1)
        feats[f"block_cls_token_{j}"] = x
    x = w.ln(x, "norm", eps)                                          # :430
    feats["features_all"] = x
    x = x[:, 0]                                                       # :432
    feats["features"] = x
    if cfg.nb_classes > 0:
        x = w.dense(x, "head")                                        # :441
    feats["logits"] = x
    return finish(x, feats, return_features)
