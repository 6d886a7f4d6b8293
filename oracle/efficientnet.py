"""fp32 CPU restatement of the reference's EfficientNet family (test infrastructure).

Follows efficientnet.py (EfficientNet.forward_features/call :278-345), efficientnet_blocks.py
(BlockArgs.decode :114-169, SqueezeExcite.call :241-248, ConvBnAct/DepthwiseSeparableConv/
InvertedResidual/EdgeResidual.call) and efficientnet_builder.py (round_channels, _scale_stage_depth,
decode_architecture, EfficientNetBuilder.__call__/_make_block).  The architecture string is decoded
here independently of the engine's own decoder.
"""
import math
import re
from collections import OrderedDict

from . import ops
from .common import BN_EPS, W, finish


def _make_divisible(value, divisor, min_value=None, round_limit=0.9):   # utils/etc.py:14-26
    min_value = min_value or divisor
    new_value = max(min_value, int(value + divisor / 2) // divisor * divisor)
    if new_value < round_limit * value:
        new_value += divisor
    return new_value


def _decode(block_string):                                              # efficientnet_blocks.py:114-169
    ops_ = block_string.split("_")
    o = {"block_type": ops_[0]}
    for op in ops_[1:]:
        if op == "noskip":
            o["skip"] = False
        elif op == "skip":
            o["skip"] = True
        elif op.startswith("n"):
            o["n"] = {"re": "relu", "r6": "relu6", "hs": "hard_swish", "sw": "swish", "mi": "mish"}[op[1:]]
        else:
            sp = re.split(r"(\d.*)", op)
            if len(sp) >= 2:
                o[sp[0]] = sp[1]
    t = o["block_type"]
    ks = lambda s: int(s) if s.isdigit() else int(s.split(".")[0])     # noqa: E731
    return dict(type=t, r=int(o["r"]), c=int(o["c"]), fc=int(o.get("fc", 0)) or None,
                exp_k=ks(o.get("a", "1")) if t != "er" else ks(o["k"]), dw_k=ks(o["k"]) if t != "er" else 1,
                s=int(o["s"]), e=float(o.get("e", 1.0)), pw_act=t == "dsa", se=float(o.get("se", 0.0)),
                act=o.get("n"), skip=False if t == "dsa" else o.get("skip", True))


def _scale(stack, mult):                                                # efficientnet_builder.py:47-93
    repeats = [b["r"] for b in stack]
    nb = sum(repeats)
    nb_scaled = int(math.ceil(nb * mult))
    out_r = []
    for r in repeats[::-1]:
        rs = max(1, round(r / nb * nb_scaled))
        out_r.append(rs)
        nb -= r
        nb_scaled -= rs
    out = []
    for b, rep in zip(stack, out_r[::-1]):
        out.extend(dict(b) for _ in range(rep))
    return out


def _se(w, x, prefix, act):                                             # SqueezeExcite.call
    s = x.mean(dim=(1, 2), keepdim=True)
    s = ops.conv2d(s, w(prefix + "/conv_reduce/kernel"), w(prefix + "/conv_reduce/bias"))
    s = ops.activation(s, act)
    s = ops.conv2d(s, w(prefix + "/conv_expand/kernel"), w(prefix + "/conv_expand/bias"))
    return x * ops.activation(s, "sigmoid")


def efficientnet_forward(cfg, weights, x, return_features=False):
    w = W(weights)
    eps = BN_EPS[cfg.norm_layer]
    x = ops.as_t(x)
    feats = OrderedDict()

    def conv(x, name, k, stride=1):                                     # create_conv2d / PadConv2D (layers/conv.py)
        kern = w(name + "/kernel")
        if cfg.padding == "symmetric":
            p = ((stride - 1) + (k - 1)) // 2
            return ops.conv2d(ops.zero_pad2d(x, p), kern, stride=stride)
        return ops.conv2d(x, kern, stride=stride, padding=cfg.padding)

    def dwconv(x, name, k, stride):
        kern = w(name + "/depthwise_kernel")
        if cfg.padding == "symmetric":
            p = ((stride - 1) + (k - 1)) // 2
            return ops.depthwise_conv2d(ops.zero_pad2d(x, p), kern, stride=stride)
        return ops.depthwise_conv2d(x, kern, stride=stride, padding=cfg.padding)

    x = conv(x, "conv_stem", 3, 2)
    x = ops.activation(w.bn(x, "bn1", eps), cfg.act_layer)
    feats["stem"] = x

    n_stacks = len(cfg.architecture)
    for si, block_strings in enumerate(cfg.architecture):
        stack = [_decode(s) for s in block_strings]
        fix = cfg.fix_first_last and si in {0, n_stacks - 1}
        stack = _scale(stack, 1.0 if fix else cfg.depth_multiplier)     # decode_architecture
        for bi, ba in enumerate(stack):
            if bi >= 1:
                ba["s"] = 1                                             # efficientnet_builder.py:258-259
            p = f"blocks.{si}.{bi}"
            filters = _make_divisible(ba["c"] * cfg.channel_multiplier, 8)       # round_channels
            act = ba["act"] or cfg.act_layer
            se_ratio = ba["se"] / ba["e"] if ba["type"] != "cn" else ba["se"]   # :201
            in_ch = x.shape[-1]
            skip = ba["s"] == 1 and filters == in_ch and ba["skip"]
            shortcut = x
            t = ba["type"]
            if t in ("ds", "dsa"):                                      # DepthwiseSeparableConv.call
                x = dwconv(x, p + "/conv_dw", ba["dw_k"], ba["s"])
                x = ops.activation(w.bn(x, p + "/bn1", eps), act)
                if se_ratio > 0:
                    x = _se(w, x, p + "/se", act)
                x = conv(x, p + "/conv_pw", 1)
                x = w.bn(x, p + "/bn2", eps)
                if ba["pw_act"]:
                    x = ops.activation(x, act)
            elif t == "ir":                                             # InvertedResidual.call
                x = conv(x, p + "/conv_pw", ba["exp_k"])
                x = ops.activation(w.bn(x, p + "/bn1", eps), act)
                x = dwconv(x, p + "/conv_dw", ba["dw_k"], ba["s"])
                x = ops.activation(w.bn(x, p + "/bn2", eps), act)
                if se_ratio > 0:
                    x = _se(w, x, p + "/se", act)
                x = conv(x, p + "/conv_pwl", 1)
                x = w.bn(x, p + "/bn3", eps)
            elif t == "er":                                             # EdgeResidual.call
                x = conv(x, p + "/conv_exp", ba["exp_k"], ba["s"])
                x = ops.activation(w.bn(x, p + "/bn1", eps), act)
                if se_ratio > 0:
                    x = _se(w, x, p + "/se", act)
                x = conv(x, p + "/conv_pwl", 1)
                x = w.bn(x, p + "/bn2", eps)
            elif t == "cn":                                             # ConvBnAct.call
                x = conv(x, p + "/conv", ba["dw_k"], ba["s"])
                x = ops.activation(w.bn(x, p + "/bn1", eps), act)
            else:
                raise ValueError(t)
            if skip:
                x = x + shortcut
            feats[f"stage_{si}/block_{bi}"] = x

    x = conv(x, "conv_head", 1)
    x = ops.activation(w.bn(x, "bn2", eps), cfg.act_layer)
    feats["conv_features"] = x
    x = ops.global_avg_pool(x)                                          # efficientnet.py:338-339
    feats["features"] = x
    if cfg.nb_classes > 0:
        x = w.dense(x, "classifier")
    feats["logits"] = x
    return finish(x, feats, return_features)
