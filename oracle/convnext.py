"""CPU restatement of reference tfimm/architectures/convnext.py (test infrastructure).

ConvNeXtBlock.call :222-232, ConvNeXtStage.call :287-298, ConvNeXt.forward_features :383-411,
ConvNeXt.call :413-445.  MLP / ConvMLP: layers/transformers.py:208-214 / 256-262 (a 1x1 conv on NHWC is
the same contraction as a Dense over the last axis).  Parity status: see oracle/ops.py.
"""
from collections import OrderedDict

from . import ops
from .common import LN_EPS, W, finish


def convnext_forward(cfg, weights, x, return_features=False):
    w = W(weights)
    eps = LN_EPS[cfg.norm_layer]
    x = ops.as_t(x)
    feats = OrderedDict()
    x = ops.conv2d(x, w("stem/0/kernel"), w("stem/0/bias"), stride=cfg.patch_size)          # :404
    x = w.ln(x, "stem/1", eps)                                                                # :405
    feats["stem"] = x
    for j, nb in enumerate(cfg.nb_blocks):
        if j > 0:                                                                             # :291-294
            x = w.ln(x, f"stages/{j}/downsample/0", eps)
            x = ops.conv2d(x, w(f"stages/{j}/downsample/1/kernel"), w(f"stages/{j}/downsample/1/bias"), stride=2)
            feats[f"stage_{j}/downsample"] = x
        for i in range(nb):
            p = f"stages/{j}/blocks/{i}"
            shortcut = x
            y = ops.zero_pad2d(x, 3)                                                          # :224
            y = ops.depthwise_conv2d(y, w(p + "/conv_dw/depthwise_kernel"), w(p + "/conv_dw/bias"))   # :225
            y = w.ln(y, p + "/norm", eps)                                                     # :226
            k1, k2 = w(p + "/mlp/fc1/kernel"), w(p + "/mlp/fc2/kernel")
            if k1.dim() == 4:                                                                 # ConvMLP: 1x1 convs
                k1, k2 = k1[0, 0], k2[0, 0]
            y = ops.activation(ops.dense(y, k1, w(p + "/mlp/fc1/bias")), cfg.act_layer)      # :227
            y = ops.dense(y, k2, w(p + "/mlp/fc2/bias"))
            y = y * w(p + "/gamma")                                                           # :228
            x = y + shortcut                                                                  # :229-230 (drop_path = id)
            feats[f"stage_{j}/block_{i}"] = x
    feats["conv_features"] = x
    conv_features = x
    x = ops.global_avg_pool(x)                                                                # :431
    x = w.ln(x, "head/norm", eps)                                                             # :432
    feats["features"] = x
    if cfg.nb_classes > 0:
        x = w.dense(x, "head/fc")                                                             # :436
    feats["logits"] = x
    return finish(x, feats, return_features)
