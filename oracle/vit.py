"""fp32 CPU restatement of reference tfimm/architectures/vit.py (test infrastructure).

Follows ViT.forward_features / ViT.call (vit.py:422-478), ViTBlock.call (:219-235),
ViTMultiHeadAttention.call (:149-171), PatchEmbeddings.call (layers/transformers.py:142-173)
and MLP.call (:208-214) statement by statement.
"""
from collections import OrderedDict

import torch

from . import ops
from .common import LN_EPS, W, finish, mlp


def _attention(w: W, x, prefix, nb_heads, qkv_bias):
    # vit.py:149-171
    B, N, D = x.shape
    scale = (D // nb_heads) ** -0.5                                   # :139-140
    qkv = w.dense(x, prefix + "/qkv", bias=qkv_bias)                  # :155  (B, N, 3D)
    qkv = qkv.reshape(B, N, 3, nb_heads, -1).permute(2, 0, 3, 1, 4)   # :156-157 (3, B, H, N, hd)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = scale * (q @ k.transpose(-1, -2))                          # :160 scale AFTER the matmul
    attn = ops.softmax(attn, -1)                                      # :161
    y = attn @ v                                                      # :165 (B, H, N, hd)
    y = y.permute(0, 2, 1, 3).reshape(B, N, -1)                       # :166-167
    return w.dense(y, prefix + "/proj"), attn                         # :169


def vit_forward(cfg, weights, x, return_features=False):
    w = W(weights)
    eps = LN_EPS[cfg.norm_layer]
    x = ops.as_t(x)
    feats = OrderedDict()
    B = x.shape[0]
    # PatchEmbeddings: ZeroPadding2D(0) -> Conv2D(k=s=patch, bias) -> flatten -> norm("") identity
    x = ops.conv2d(x, w("patch_embed/proj/kernel"), w("patch_embed/proj/bias"), stride=cfg.patch_size)
    _grid = (x.shape[1], x.shape[2])                                  # return_shape=True, transformers.py:171-172
    x = x.reshape(B, -1, x.shape[-1])                                 # transformers.py:167-170
    cls = w("cls_token").expand(B, -1, -1)                            # vit.py:427 tf.repeat
    if not cfg.distilled:
        x = torch.cat((cls, x), dim=1)                                # :429
    else:
        dist = w("dist_token").expand(B, -1, -1)
        x = torch.cat((cls, dist, x), dim=1)                          # :431-432
    pos = w("pos_embed")
    if cfg.interpolate_input:                                         # :433-442
        grid = _grid
        pos = ops.interpolate_pos_embeddings(pos, cfg.grid_size, grid, cfg.nb_tokens)
    x = x + pos                                                       # :434
    feats["patch_embedding"] = x
    for j in range(cfg.nb_blocks):
        p = f"blocks/{j}"
        shortcut = x
        y = w.ln(x, p + "/norm1", eps)                                # :222
        y, attn = _attention(w, y, p + "/attn", cfg.nb_heads, cfg.qkv_bias)
        x = y + shortcut                                              # :227-228 (drop_path identity)
        shortcut = x
        y = w.ln(x, p + "/norm2", eps)                                # :231
        y = mlp(w, y, p + "/mlp", cfg.act_layer)                      # :232
        x = y + shortcut                                              # :234
        feats[f"block_{j}/attn"] = attn
        feats[f"block_{j}"] = x
    x = w.ln(x, "norm", eps)                                          # :452
    feats["features_all"] = x
    if cfg.distilled:
        x = x[:, :2]                                                  # :458
    elif cfg.representation_size:
        x = torch.tanh(w.dense(x[:, 0], "pre_logits/fc"))             # :460, :351-359
    else:
        x = x[:, 0]                                                   # :462
    feats["features"] = x
    if cfg.nb_classes > 0:
        if not cfg.distilled:
            x = w.dense(x, "head")                                    # :472
        else:
            y = w.dense(x[:, 0], "head")                              # :474
            yd = w.dense(x[:, 1], "head_dist")                        # :475
            x = torch.stack((y, yd), dim=1)                           # :476
    feats["logits"] = x
    return finish(x, feats, return_features)
