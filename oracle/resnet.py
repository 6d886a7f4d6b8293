"""fp32 CPU restatement of reference tfimm/architectures/resnet.py (test infrastructure).

Follows ResNet.forward_features / call (resnet.py:570-593), BasicBlock.call (:166-189),
Bottleneck.call (:266-292), downsample_conv / downsample_avg (:295-330), make_stage
(:333-382), SEModule.call (layers/attention.py:66-74) and ClassifierHead.call
(layers/classifier.py:65-74).  Convs and BNs are kept as separate steps, as in the reference.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ops
from .common import BN_EPS, W, finish


def _se(w: W, x, prefix):
    # layers/attention.py:66-74 (bn = identity, act = relu, gate = sigmoid)
    s = x.mean(dim=(1, 2), keepdim=True)
    s = ops.conv2d(s, w(prefix + "/fc1/kernel"), w(prefix + "/fc1/bias"))
    s = ops.activation(s, "relu")
    s = ops.conv2d(s, w(prefix + "/fc2/kernel"), w(prefix + "/fc2/bias"))
    return x * ops.activation(s, "sigmoid")


def _eca(w: W, x, prefix):
    """EcaModule.call (layers/attention.py:120-130): channel means -> zero-padded Conv1D over the channel axis
    (one filter, no bias; kernel size from the channel count, :105-110) -> sigmoid -> scale."""
    k = w(prefix + "/conv/kernel").reshape(-1)                          # (k, 1, 1)
    pad = (k.shape[0] - 1) // 2
    y = x.mean(dim=(1, 2))                                              # (N, C)
    y = F.conv1d(F.pad(y[:, None, :], (pad, pad)), k.reshape(1, 1, -1))[:, 0, :]
    return x * torch.sigmoid(y)[:, None, None, :]


def _norm(w: W, cfg):
    """norm_layer_factory (layers/factory.py:16-60): BatchNormalization, or GroupNormalization with its defaults
    (32 groups, eps 1e-5; layers/norm.py:128-139)."""
    if cfg.norm_layer == "group_norm":
        return lambda x, prefix: ops.group_norm(x, w(prefix + "/gamma"), w(prefix + "/beta"), 32, 1e-5)
    eps = BN_EPS[cfg.norm_layer]
    return lambda x, prefix: w.bn(x, prefix, eps)


def _downsample(w: W, cfg, x, prefix, stride, norm):
    if cfg.downsample_mode == "avg":                                   # resnet.py:295-312
        if stride != 1:
            x = ops.avg_pool2d_same(x, 2, stride)
        x = ops.conv2d(x, w(prefix + "/downsample/1/kernel"))
        return norm(x, prefix + "/downsample/2")
    p = (stride + cfg.down_kernel_size) // 2 - 1                       # resnet.py:319
    x = ops.zero_pad2d(x, p)
    x = ops.conv2d(x, w(prefix + "/downsample/0/kernel"), stride=stride)
    return norm(x, prefix + "/downsample/1")


def resnet_forward(cfg, weights, x, return_features=False):
    assert cfg.aa_layer in ("", "blur_pool") and cfg.attn_layer in ("", "se", "eca")
    w = W(weights)
    norm = _norm(w, cfg)
    act = cfg.act_layer
    x = ops.as_t(x)
    feats = OrderedDict()
    # ---- stem (resnet.py:572-576)
    if cfg.stem_type in ("deep", "deep_tiered"):
        x = ops.zero_pad2d(x, 1)                                        # pad1
        x = ops.conv2d(x, w("conv1/0/kernel"), stride=2)
        x = ops.activation(norm(x, "conv1/1"), act)
        x = ops.conv2d(x, w("conv1/3/kernel"), padding="same")
        x = ops.activation(norm(x, "conv1/4"), act)
        x = ops.conv2d(x, w("conv1/6/kernel"), padding="same")
    else:
        x = ops.zero_pad2d(x, 3)
        x = ops.conv2d(x, w("conv1/kernel"), stride=2)
    x = ops.activation(norm(x, "bn1"), act)
    if cfg.replace_stem_pool:                                           # resnet.py:517-530
        x = ops.zero_pad2d(x, 1)
        x = ops.conv2d(x, w("maxpool/0/kernel"), stride=2)
        x = ops.activation(norm(x, "maxpool/1"), act)
    elif cfg.aa_layer:                                                  # resnet.py:532-536
        x = ops.max_pool2d(ops.zero_pad2d(x, 1), 3, 1)
        x = ops.blur_pool2d(x, 2)
    else:                                                               # resnet.py:537-540
        x = ops.max_pool2d(ops.zero_pad2d(x, 1), 3, 2)
    feats["stem"] = x
    # ---- stages (make_stage, resnet.py:333-382)
    expansion = 1 if cfg.block == "basic_block" else 4
    in_channels = cfg.stem_width * 2 if cfg.stem_type in ("deep", "deep_tiered") else 64
    j = 0
    for idx in range(4):
        nb_channels = cfg.nb_channels[idx]
        out_channels = nb_channels * expansion
        for block_idx in range(cfg.nb_blocks[idx]):
            stride = 1 if idx == 0 or block_idx > 0 else 2
            p = f"layer{idx + 1}/{block_idx}"
            has_down = block_idx == 0 and (stride != 1 or in_channels != out_channels)
            shortcut = x
            use_aa = bool(cfg.aa_layer) and stride == 2                 # resnet.py:127, 218
            cstride = 1 if use_aa else stride                           # the blur layer takes care of the stride
            if cfg.block == "basic_block":                              # BasicBlock.call
                y = ops.zero_pad2d(x, 1)
                y = ops.conv2d(y, w(p + "/conv1/kernel"), stride=cstride)
                y = ops.activation(norm(y, p + "/bn1"), act)
                if use_aa:
                    y = ops.blur_pool2d(y, stride)                      # :173-174
                y = ops.zero_pad2d(y, 1)
                y = ops.conv2d(y, w(p + "/conv2/kernel"))
                y = norm(y, p + "/bn2")
            else:                                                       # Bottleneck.call
                y = ops.conv2d(x, w(p + "/conv1/kernel"))
                y = ops.activation(norm(y, p + "/bn1"), act)
                y = ops.zero_pad2d(y, 1)
                y = ops.conv2d(y, w(p + "/conv2/kernel"), stride=cstride, groups=cfg.cardinality)   # resnet.py:229-236
                y = ops.activation(norm(y, p + "/bn2"), act)
                if use_aa:
                    y = ops.blur_pool2d(y, stride)                      # :277-278
                y = ops.conv2d(y, w(p + "/conv3/kernel"))
                y = norm(y, p + "/bn3")
            if cfg.attn_layer == "se":
                y = _se(w, y, p + "/se")
            elif cfg.attn_layer == "eca":
                y = _eca(w, y, p + "/se")
            if has_down:
                shortcut = _downsample(w, cfg, shortcut, p, stride, norm)
            x = ops.activation(y + shortcut, act)
            feats[f"block_{j}"] = x
            j += 1
            in_channels = nb_channels                                   # resnet.py:380
    feats["features"] = x
    # ---- head: GAP -> fc -> flatten (layers/classifier.py:65-74)
    x = ops.global_avg_pool(x)
    if cfg.nb_classes > 0:
        x = w.dense(x, "remove/fc")
    feats["logits"] = x
    return finish(x, feats, return_features)
