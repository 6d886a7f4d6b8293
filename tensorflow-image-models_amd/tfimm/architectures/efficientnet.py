"""EfficientNet / EfficientNet-V2 / -Lite / -EdgeTPU / MobileNet-V2 on the MI355X engine.

Behavioural mirror of the reference's three files:
  tfimm/architectures/efficientnet.py          EfficientNetConfig :119-190, EfficientNet :193-345, cfg helpers
  tfimm/architectures/efficientnet_blocks.py   BlockArgs.decode :114-169, SqueezeExcite :189-248, ConvBnAct,
                                               DepthwiseSeparableConv, InvertedResidual, EdgeResidual :251-535
  tfimm/architectures/efficientnet_builder.py  round_channels :31-44, _scale_stage_depth :47-93,
                                               decode_architecture :96-144, EfficientNetBuilder :147-283

Lowering of an MBConv block (InvertedResidual.call, efficientnet_blocks.py:438-453):
  expand 1x1 + BN + act          -> one GEMM launch (folded BN, activation epilogue)
  depthwise kxk + BN + act       -> one depthwise launch that ALSO accumulates the per-(image, channel)
                                    sums the SqueezeExcite squeeze needs (no extra pass over the tensor)
  SE gate (2 tiny FCs + sigmoid) -> one launch, one workgroup per image
  project 1x1 + BN (+ shortcut)  -> one GEMM launch whose A-operand loader multiplies by the SE gate
                                    (the `x * gate` tensor is never materialised) + residual epilogue
"""
import math
import re
from collections import OrderedDict
from copy import deepcopy
from dataclasses import dataclass
from typing import List, Optional, Tuple

from ..models.config import ModelConfig
from ..models.model import Model, WeightSpec
from ..models.registry import register_model
from ..utils.constants import (
    IMAGENET_DEFAULT_MEAN,
    IMAGENET_DEFAULT_STD,
    IMAGENET_INCEPTION_MEAN,
    IMAGENET_INCEPTION_STD,
)
from ..utils.etc import get_padding, make_divisible

__all__ = ["EfficientNet", "EfficientNetConfig"]

_BN_EPS = {"batch_norm": 1e-5, "batch_norm_tf": 1e-3}


@dataclass
class EfficientNetConfig(ModelConfig):
    nb_classes: int = 1000
    in_channels: int = 3
    input_size: Tuple[int, int] = (224, 224)
    # Architecture
    stem_size: int = 32
    architecture: Tuple[Tuple[str, ...], ...] = ()
    channel_multiplier: float = 1.0
    depth_multiplier: float = 1.0
    fix_first_last: bool = False
    nb_features: int = 1280
    # Regularization
    drop_rate: float = 0.0
    drop_path_rate: float = 0.0
    # Other params
    norm_layer: str = "batch_norm"
    act_layer: str = "swish"
    padding: str = "symmetric"  # "symmetric" | "same" | "valid"
    # Parameters for inference
    crop_pct: float = 0.875
    interpolation: str = "bicubic"
    # Preprocessing
    mean: Tuple[float, float, float] = IMAGENET_DEFAULT_MEAN
    std: Tuple[float, float, float] = IMAGENET_DEFAULT_STD
    # Weight transfer
    first_conv: str = "conv_stem"
    classifier: str = "classifier"


# ---------------------------------------------------------------------------------------
# block DSL + scaling arithmetic
# ---------------------------------------------------------------------------------------
@dataclass
class BlockArgs:
    block_type: str
    nb_repeats: int
    filters: int
    force_in_channels: Optional[int]
    exp_kernel: int
    dw_kernel: int
    pw_kernel: int
    stride: int
    exp_ratio: float
    pw_act: bool
    se_ratio: float
    act_layer: Optional[str]
    skip: bool


_ACT_CODES = {"re": "relu", "r6": "relu6", "hs": "hard_swish", "sw": "swish", "mi": "mish"}


def _ksize(s: str) -> int:
    if s.isdigit():
        return int(s)
    a, b = s.split(".")
    assert a == b, "non-square kernels are not used by any registered model"
    return int(a)


def decode_block(block_string: str) -> BlockArgs:
    """``ir_r2_k3_s2_e6_c24_se0.25`` -> BlockArgs (efficientnet_blocks.py:114-169)."""
    ops = block_string.split("_")
    opt = {}
    skip = None
    act = None
    for op in ops[1:]:
        if op == "noskip":
            skip = False
        elif op == "skip":
            skip = True
        elif op.startswith("n"):
            act = _ACT_CODES[op[1:]]
        else:
            parts = re.split(r"(\d.*)", op)
            if len(parts) >= 2:
                opt[parts[0]] = parts[1]
    btype = ops[0]
    if skip is None:
        skip = True
    if btype == "dsa":
        skip = False
    if btype != "er":
        exp_k, dw_k = _ksize(opt.get("a", "1")), _ksize(opt["k"])
    else:
        exp_k, dw_k = _ksize(opt["k"]), 1
    assert "cc" not in opt and "gs" not in opt, "CondConv / grouped blocks are not used by any registered model"
    return BlockArgs(block_type=btype, nb_repeats=int(opt["r"]), filters=int(opt["c"]),
                     force_in_channels=int(opt["fc"]) if "fc" in opt else None, exp_kernel=exp_k, dw_kernel=dw_k,
                     pw_kernel=_ksize(opt.get("p", "1")), stride=int(opt["s"]), exp_ratio=float(opt.get("e", 1.0)),
                     pw_act=btype == "dsa", se_ratio=float(opt.get("se", 0.0)), act_layer=act, skip=skip)


def round_channels(channels, multiplier=1.0, divisor=8, min_channels=None, round_limit=0.9):
    return make_divisible(channels * multiplier, divisor, min_channels, round_limit)


def _scale_stage_depth(stack: List[BlockArgs], multiplier: float) -> List[BlockArgs]:
    """Per-stage "ceil" depth scaling, distributed back to front (efficientnet_builder.py:47-93)."""
    repeats = [ba.nb_repeats for ba in stack]
    total = sum(repeats)
    scaled_total = int(math.ceil(total * multiplier))
    scaled = []
    for r in repeats[::-1]:
        rs = max(1, round(r / total * scaled_total))
        scaled.append(rs)
        total -= r
        scaled_total -= rs
    scaled = scaled[::-1]
    out = []
    for ba, rep in zip(stack, scaled):
        out.extend(deepcopy(ba) for _ in range(rep))
    return out


@dataclass
class BlockPlan:
    """One concrete block after scaling: everything lower() / weight_specs() need."""
    name: str            # "blocks.{stage}.{idx}"
    key: str             # feature key "stage_{s}/block_{b}"
    type: str
    cin: int
    mid: int
    cout: int
    k: int               # dw kernel (ds/ir), expansion kernel (er), conv kernel (cn)
    pw_k: int
    stride: int
    act: str
    pw_act: bool
    rd: int              # SE reduction channels, 0 = no SE
    skip: bool


def plan_blocks(cfg: EfficientNetConfig) -> List[BlockPlan]:
    arch = []
    for si, block_strings in enumerate(cfg.architecture):
        stack = [decode_block(s) for s in block_strings]
        fix = cfg.fix_first_last and si in {0, len(cfg.architecture) - 1}
        arch.append(_scale_stage_depth(stack, 1.0 if fix else cfg.depth_multiplier))
    plans: List[BlockPlan] = []
    cin = cfg.stem_size
    current_stride = 2
    for si, stack in enumerate(arch):
        for bi, ba in enumerate(stack):
            stride = ba.stride if bi == 0 else 1            # only the first block of a stack strides
            if stride > 1:
                current_stride *= stride
                assert current_stride <= 32, "dilated variants (output_stride < total stride) are not built"
            cout = round_channels(ba.filters, cfg.channel_multiplier)
            act = ba.act_layer or cfg.act_layer
            se_ratio = ba.se_ratio
            if ba.block_type != "cn":
                se_ratio = se_ratio / ba.exp_ratio           # efficientnet_builder.py:201
            t = ba.block_type
            if t in ("ds", "dsa"):
                mid, k = cin, ba.dw_kernel
            elif t == "ir":
                mid, k = make_divisible(cin * ba.exp_ratio, 8), ba.dw_kernel
            elif t == "er":
                fc = ba.force_in_channels
                fc = round_channels(fc, cfg.channel_multiplier) if fc is not None else cin
                mid, k = make_divisible(fc * ba.exp_ratio, 8), ba.exp_kernel
            elif t == "cn":
                mid, k = cout, ba.dw_kernel
            else:
                raise ValueError(f"Unknown block type {t} while building model.")
            rd = 0
            if t != "cn" and se_ratio > 0.0:
                rd = round(mid * se_ratio)                   # efficientnet_blocks.py:225 (python round)
            skip = stride == 1 and cout == cin and ba.skip
            plans.append(BlockPlan(f"blocks.{si}.{bi}", f"stage_{si}/block_{bi}", "ds" if t == "dsa" else t, cin, mid,
                                   cout, k, ba.pw_kernel, stride, act, ba.pw_act, rd, skip))
            cin = cout
    return plans


def _bn(s, prefix, c, last_of_residual=False):
    # "small": hint for the synthetic weight generator only (keeps deep residual stacks well scaled)
    s[prefix + "/gamma"] = WeightSpec((c,), "gamma", "small" if last_of_residual else "")
    s[prefix + "/beta"] = WeightSpec((c,), "beta")
    s[prefix + "/moving_mean"] = WeightSpec((c,), "mean")
    s[prefix + "/moving_variance"] = WeightSpec((c,), "var")


def _se(s, prefix, c, rd):
    s[prefix + "/conv_reduce/kernel"] = WeightSpec((1, 1, c, rd), "conv")
    s[prefix + "/conv_reduce/bias"] = WeightSpec((rd,), "bias")
    s[prefix + "/conv_expand/kernel"] = WeightSpec((1, 1, rd, c), "conv")
    s[prefix + "/conv_expand/bias"] = WeightSpec((c,), "bias")


class EfficientNet(Model):
    cfg_class = EfficientNetConfig

    def __init__(self, cfg, *args, **kwargs):
        self._block_plans = None
        super().__init__(cfg, *args, **kwargs)

    def block_plans(self) -> List[BlockPlan]:
        if self._block_plans is None:
            self._block_plans = plan_blocks(self.cfg)
        return self._block_plans

    # ---- variables -----------------------------------------------------------------
    def weight_specs(self):
        c = self.cfg
        s = OrderedDict()
        s["conv_stem/kernel"] = WeightSpec((3, 3, c.in_channels, c.stem_size), "conv")
        _bn(s, "bn1", c.stem_size)
        last = c.stem_size
        for b in self.block_plans():
            p = b.name
            if b.type == "ds":
                s[p + "/conv_dw/depthwise_kernel"] = WeightSpec((b.k, b.k, b.cin, 1), "dwconv")
                _bn(s, p + "/bn1", b.cin)
                if b.rd:
                    _se(s, p + "/se", b.cin, b.rd)
                s[p + "/conv_pw/kernel"] = WeightSpec((b.pw_k, b.pw_k, b.cin, b.cout), "conv")
                _bn(s, p + "/bn2", b.cout, b.skip)
            elif b.type == "ir":
                s[p + "/conv_pw/kernel"] = WeightSpec((1, 1, b.cin, b.mid), "conv")
                _bn(s, p + "/bn1", b.mid)
                s[p + "/conv_dw/depthwise_kernel"] = WeightSpec((b.k, b.k, b.mid, 1), "dwconv")
                _bn(s, p + "/bn2", b.mid)
                if b.rd:
                    _se(s, p + "/se", b.mid, b.rd)
                s[p + "/conv_pwl/kernel"] = WeightSpec((b.pw_k, b.pw_k, b.mid, b.cout), "conv")
                _bn(s, p + "/bn3", b.cout, b.skip)
            elif b.type == "er":
                s[p + "/conv_exp/kernel"] = WeightSpec((b.k, b.k, b.cin, b.mid), "conv")
                _bn(s, p + "/bn1", b.mid)
                if b.rd:
                    _se(s, p + "/se", b.mid, b.rd)
                s[p + "/conv_pwl/kernel"] = WeightSpec((b.pw_k, b.pw_k, b.mid, b.cout), "conv")
                _bn(s, p + "/bn2", b.cout, b.skip)
            else:  # cn
                s[p + "/conv/kernel"] = WeightSpec((b.k, b.k, b.cin, b.cout), "conv")
                _bn(s, p + "/bn1", b.cout, b.skip)
            last = b.cout
        s["conv_head/kernel"] = WeightSpec((1, 1, last, c.nb_features), "conv")
        _bn(s, "bn2", c.nb_features)
        if c.nb_classes > 0:
            s["classifier/kernel"] = WeightSpec((c.nb_features, c.nb_classes), "dense")
            s["classifier/bias"] = WeightSpec((c.nb_classes,), "bias")
        return s

    @property
    def feature_names(self) -> List[str]:
        return ["stem"] + [b.key for b in self.block_plans()] + ["conv_features", "features", "logits"]

    # ---- lowering ---------------------------------------------------------------------------------------
    def lower(self, b, H, W, want_features):
        c = self.cfg
        if c.norm_layer not in _BN_EPS:
            raise NotImplementedError(f"norm_layer={c.norm_layer!r} is not built.")
        eps = _BN_EPS[c.norm_layer]

        def pad(k, s):
            if c.padding == "same":
                return "same"
            if c.padding == "symmetric":
                return get_padding(k, s)[0]          # layers/conv.py:24-27
            return 0

        x = b.image_input(H, W, c.in_channels)
        plans = self.block_plans()
        # stem + the first block's depthwise layer as one launch when nothing else reads the stem output: a depthwise-separable
        # first block that takes no shortcut from the stem, and no "stem" feature requested
        stem_fused = None
        first = plans[0] if plans else None
        if (first is not None and first.type == "ds" and first.k == 3 and first.stride == 1 and not first.skip
                and not want_features and c.act_layer == first.act):
            stem_fused = b.stem_dwconv(x, "conv_stem/kernel", "bn1", first.name + "/conv_dw/depthwise_kernel",
                                       first.name + "/bn1", bn_eps=eps, padding=pad(3, 2), dw_padding=pad(3, 1),
                                       act=c.act_layer, squeeze=first.rd > 0,
                                       cite="efficientnet.py:300-302, efficientnet_blocks.py:350-352")
        if stem_fused is None:
            x = b.conv(x, "conv_stem/kernel", stride=2, padding=pad(3, 2), bn="bn1", bn_eps=eps, act=c.act_layer,
                       cite="efficientnet.py:300-302")
        if want_features:
            b.p.mark_output("stem", x)
        for blk in plans:
            p = blk.name
            shortcut = x if blk.skip else None
            if blk.type in ("ds", "ir"):
                fused = stem_fused if blk is first else None
                if blk.type == "ir":
                    dw_bn, proj, proj_bn, proj_act = p + "/bn2", p + "/conv_pwl/kernel", p + "/bn3", ""
                    # narrow inputs (the first stages): expansion + depthwise as one launch, the expanded tensor stays in LDS
                    fused = b.expand_dwconv(x, p + "/conv_pw/kernel", p + "/bn1", p + "/conv_dw/depthwise_kernel", dw_bn,
                                            bn_eps=eps, stride=blk.stride, padding=pad(blk.k, blk.stride), act=blk.act,
                                            squeeze=blk.rd > 0, cite="efficientnet_blocks.py:438-445")
                    if fused is None:
                        x = b.conv(x, p + "/conv_pw/kernel", bn=p + "/bn1", bn_eps=eps, act=blk.act,
                                   cite="efficientnet_blocks.py:440-442")
                else:
                    dw_bn, proj, proj_bn = p + "/bn1", p + "/conv_pw/kernel", p + "/bn2"
                    proj_act = blk.act if blk.pw_act else ""
                if fused is not None:
                    x, sums = fused
                else:
                    x, sums = b.dwconv(x, p + "/conv_dw/depthwise_kernel", stride=blk.stride, padding=pad(blk.k, blk.stride),
                                       bn=dw_bn, bn_eps=eps, act=blk.act, squeeze=blk.rd > 0,
                                       cite="efficientnet_blocks.py:350-352,443-445")
                gate = None
                if blk.rd:
                    gate = b.se_gate(sums, x.rows, p + "/se/conv_reduce/kernel", p + "/se/conv_reduce/bias",
                                     p + "/se/conv_expand/kernel", p + "/se/conv_expand/bias", act=blk.act,
                                     cite="efficientnet_blocks.py:241-247")
                x = self._project(b, x, proj, proj_bn, eps, proj_act, gate, shortcut, blk)
            elif blk.type == "er":
                x = b.conv(x, p + "/conv_exp/kernel", stride=blk.stride, padding=pad(blk.k, blk.stride), bn=p + "/bn1",
                           bn_eps=eps, act=blk.act, cite="efficientnet_blocks.py:522-524")
                gate = None
                if blk.rd:
                    m = b.mean_rows(x, out_f32=True, cite="efficientnet_blocks.py:242")
                    gate = b.se_gate(m, 1, p + "/se/conv_reduce/kernel", p + "/se/conv_reduce/bias",
                                     p + "/se/conv_expand/kernel", p + "/se/conv_expand/bias", act=blk.act,
                                     cite="efficientnet_blocks.py:243-247")
                x = self._project(b, x, p + "/conv_pwl/kernel", p + "/bn2", eps, "", gate, shortcut, blk)
            else:  # cn: act(bn(conv(x))) [+ x]
                x = b.conv(x, p + "/conv/kernel", stride=blk.stride, padding=pad(blk.k, blk.stride), bn=p + "/bn1",
                           bn_eps=eps, act=blk.act, residual=shortcut, act_after_res=False,
                           cite="efficientnet_blocks.py:283-292")
            if want_features:
                b.p.mark_output(blk.key, x)
        x = b.conv(x, "conv_head/kernel", bn="bn2", bn_eps=eps, act=c.act_layer, cite="efficientnet.py:309-311")
        if want_features:
            b.p.mark_output("conv_features", x)
        pooled = b.mean_rows(x, cite="efficientnet.py:338-339")
        b.p.mark_output("features", pooled)
        if c.nb_classes > 0:
            logits = b.dense(pooled, "classifier/kernel", "classifier/bias", out_f32=True, cite="efficientnet.py:343")
        else:
            logits = pooled
        b.p.mark_output("logits", logits)

    def _project(self, b, x, kernel, bn, eps, act, gate, shortcut, blk):
        """Pointwise-linear projection + BN (+ SE gate on its input, + shortcut)."""
        if blk.pw_k != 1:
            raise NotImplementedError("non-1x1 projection kernels are not used by any registered model")
        if gate is not None and x.C % 8 != 0:
            # odd channel counts: gate cannot ride in the GEMM loader -> separate scaling pass
            x = b.scale_channels(x, gate, cite="efficientnet_blocks.py:248")
            gate = None
        return b.conv(x, kernel, bn=bn, bn_eps=eps, act=act, residual=shortcut, a_scale=gate,
                      cite="efficientnet_blocks.py:355-361,448-452")


# ---------------------------------------------------------------------------------------
# registrations (reference efficientnet.py:348-1640)
# ---------------------------------------------------------------------------------------
_ARCH_B = (("ds_r1_k3_s1_e1_c16_se0.25",), ("ir_r2_k3_s2_e6_c24_se0.25",), ("ir_r2_k5_s2_e6_c40_se0.25",),
           ("ir_r3_k3_s2_e6_c80_se0.25",), ("ir_r3_k5_s1_e6_c112_se0.25",), ("ir_r4_k5_s2_e6_c192_se0.25",),
           ("ir_r1_k3_s1_e6_c320_se0.25",))
_ARCH_LITE = tuple((s[0].replace("_se0.25", ""),) for s in _ARCH_B)
_ARCH_EDGE = (("er_r1_k3_s1_e4_c24_fc24_noskip",), ("er_r2_k3_s2_e8_c32",), ("er_r4_k3_s2_e8_c48",),
              ("ir_r5_k5_s2_e8_c96",), ("ir_r4_k5_s1_e8_c144",), ("ir_r2_k5_s2_e8_c192",))
_ARCH_MNV2 = (("ds_r1_k3_s1_c16",), ("ir_r2_k3_s2_e6_c24",), ("ir_r3_k3_s2_e6_c32",), ("ir_r4_k3_s2_e6_c64",),
              ("ir_r3_k3_s1_e6_c96",), ("ir_r3_k3_s2_e6_c160",), ("ir_r1_k3_s1_e6_c320",))
_ARCH_V2B = (("cn_r1_k3_s1_e1_c16_skip",), ("er_r2_k3_s2_e4_c32",), ("er_r2_k3_s2_e4_c48",),
             ("ir_r3_k3_s2_e4_c96_se0.25",), ("ir_r5_k3_s1_e6_c112_se0.25",), ("ir_r8_k3_s2_e6_c192_se0.25",))
_ARCH_V2 = {
    "s": (("cn_r2_k3_s1_e1_c24_skip",), ("er_r4_k3_s2_e4_c48",), ("er_r4_k3_s2_e4_c64",),
          ("ir_r6_k3_s2_e4_c128_se0.25",), ("ir_r9_k3_s1_e6_c160_se0.25",), ("ir_r15_k3_s2_e6_c256_se0.25",)),
    "m": (("cn_r3_k3_s1_e1_c24_skip",), ("er_r5_k3_s2_e4_c48",), ("er_r5_k3_s2_e4_c80",),
          ("ir_r7_k3_s2_e4_c160_se0.25",), ("ir_r14_k3_s1_e6_c176_se0.25",), ("ir_r18_k3_s2_e6_c304_se0.25",),
          ("ir_r5_k3_s1_e6_c512_se0.25",)),
    "l": (("cn_r4_k3_s1_e1_c32_skip",), ("er_r7_k3_s2_e4_c64",), ("er_r7_k3_s2_e4_c96",),
          ("ir_r10_k3_s2_e4_c192_se0.25",), ("ir_r19_k3_s1_e6_c224_se0.25",), ("ir_r25_k3_s2_e6_c384_se0.25",),
          ("ir_r7_k3_s1_e6_c640_se0.25",)),
    "xl": (("cn_r4_k3_s1_e1_c32_skip",), ("er_r8_k3_s2_e4_c64",), ("er_r8_k3_s2_e4_c96",),
           ("ir_r16_k3_s2_e4_c192_se0.25",), ("ir_r24_k3_s1_e6_c256_se0.25",), ("ir_r32_k3_s2_e6_c512_se0.25",),
           ("ir_r8_k3_s1_e6_c640_se0.25",)),
}
# variant -> (channel multiplier, depth multiplier, drop rate)
_SCALING = {"b0": (1.0, 1.0, 0.2), "b1": (1.0, 1.1, 0.2), "b2": (1.1, 1.2, 0.3), "b3": (1.2, 1.4, 0.3),
            "b4": (1.4, 1.8, 0.4), "b5": (1.6, 2.2, 0.4), "b6": (1.8, 2.6, 0.5), "b7": (2.0, 3.1, 0.5),
            "b8": (2.2, 3.6, 0.5), "l2": (4.3, 5.3, 0.5)}
_INC = dict(mean=IMAGENET_INCEPTION_MEAN, std=IMAGENET_INCEPTION_STD)
_TF = dict(norm_layer="batch_norm_tf", padding="same")


def _effnet(name, timm_name, variant, res, crop, tf=True, **kw):
    cm, dm, dr = _SCALING[variant]
    base = dict(name=name, url="[timm]" + timm_name, input_size=(res, res), stem_size=round_channels(32, cm),
                architecture=_ARCH_B, channel_multiplier=cm, depth_multiplier=dm,
                nb_features=round_channels(1280, cm), drop_rate=dr, drop_path_rate=dr, act_layer="swish",
                crop_pct=crop, **(_TF if tf else dict(norm_layer="batch_norm", padding="symmetric")))
    base.update(kw)
    return EfficientNetConfig(**base)


def _variant(name, timm_name, arch, res, crop, cm=1.0, dm=1.0, dr=0.2, stem=32, feat=1280, act="swish", **kw):
    base = dict(name=name, url="[timm]" + timm_name, input_size=(res, res), stem_size=stem, architecture=arch,
                channel_multiplier=cm, depth_multiplier=dm, nb_features=feat, drop_rate=dr, drop_path_rate=dr,
                act_layer=act, crop_pct=crop, **_TF)
    base.update(kw)
    return EfficientNetConfig(**base)


_CFGS = []
_RES = {"b0": (224, 0.875), "b1": (240, 0.882), "b2": (260, 0.890), "b3": (300, 0.904), "b4": (380, 0.922),
        "b5": (456, 0.934), "b6": (528, 0.942), "b7": (600, 0.949), "b8": (672, 0.954)}
for _v, (_r, _c) in _RES.items():
    _CFGS.append(_effnet(f"efficientnet_{_v}", f"tf_efficientnet_{_v}", _v, _r, _c))
    _CFGS.append(_effnet(f"efficientnet_{_v}_ap", f"tf_efficientnet_{_v}_ap", _v, _r, _c, **_INC))
    if _v != "b8":
        _CFGS.append(_effnet(f"efficientnet_{_v}_ns", f"tf_efficientnet_{_v}_ns", _v, _r, _c))
_CFGS += [
    _effnet("efficientnet_l2_ns_475", "tf_efficientnet_l2_ns_475", "l2", 475, 0.936),
    _effnet("efficientnet_l2_ns", "tf_efficientnet_l2_ns", "l2", 800, 0.96),
    _effnet("pt_efficientnet_b0", "efficientnet_b0", "b0", 224, 0.875, tf=False),
    _effnet("pt_efficientnet_b1", "efficientnet_b1", "b1", 256, 1.0, tf=False),
    _effnet("pt_efficientnet_b2", "efficientnet_b2", "b2", 256, 1.0, tf=False),
    _effnet("pt_efficientnet_b3", "efficientnet_b3", "b3", 288, 1.0, tf=False),
    _effnet("pt_efficientnet_b4", "efficientnet_b4", "b4", 320, 1.0, tf=False),
    # EdgeTPU (efficientnet.py: _efficientnet_edge_cfg)
    _variant("efficientnet_es", "tf_efficientnet_es", _ARCH_EDGE, 224, 0.875, 1.0, 1.0, 0.2, act="relu", **_INC),
    _variant("efficientnet_em", "tf_efficientnet_em", _ARCH_EDGE, 240, 0.882, 1.0, 1.1, 0.2, act="relu", **_INC),
    _variant("efficientnet_el", "tf_efficientnet_el", _ARCH_EDGE, 300, 0.904, 1.2, 1.4, 0.3, stem=40, feat=1536,
             act="relu", **_INC),
]
# Lite: fixed stem/head width and first/last depth, relu6, no SE
for _n, (_r, _c, _cm, _dm, _dr) in {"lite0": (224, 0.875, 1.0, 1.0, 0.2), "lite1": (240, 0.882, 1.0, 1.1, 0.2),
                                    "lite2": (260, 0.890, 1.1, 1.2, 0.3), "lite3": (280, 0.904, 1.2, 1.4, 0.3),
                                    "lite4": (300, 0.920, 1.4, 1.8, 0.3)}.items():
    _CFGS.append(_variant(f"efficientnet_{_n}", f"tf_efficientnet_{_n}", _ARCH_LITE, _r, _c, _cm, _dm, _dr,
                          act="relu6", fix_first_last=True, **_INC))
# V2 base models: scaled from b0
for _n, (_r, _c, _v) in {"b0": (192, 0.875, "b0"), "b1": (192, 0.882, "b1"), "b2": (208, 0.890, "b2"),
                         "b3": (240, 0.904, "b3")}.items():
    _cm, _dm, _dr = _SCALING[_v]
    _CFGS.append(_variant(f"efficientnet_v2_{_n}", f"tf_efficientnetv2_{_n}", _ARCH_V2B, _r, _c, _cm, _dm, _dr,
                          stem=round_channels(32, _cm), feat=round_channels(1280, _cm)))
for _n, (_r, _dr, _stem) in {"s": (300, 0.3, 24), "m": (384, 0.4, 24), "l": (384, 0.5, 32), "xl": (384, 0.5, 32)}.items():
    for _suffix in ("", "_in21ft1k", "_in21k"):
        if _n == "xl" and _suffix == "":
            continue
        _CFGS.append(_variant(f"efficientnet_v2_{_n}{_suffix}", f"tf_efficientnetv2_{_n}{_suffix}", _ARCH_V2[_n], _r, 1.0,
                              dr=_dr, stem=_stem, nb_classes=21843 if _suffix == "_in21k" else 1000, **_INC))
# MobileNet-V2
for _n, (_cm, _dm, _fix) in {"050": (0.5, 1.0, False), "100": (1.0, 1.0, False), "140": (1.4, 1.0, False),
                             "110d": (1.1, 1.2, True), "120d": (1.2, 1.4, True)}.items():
    _CFGS.append(EfficientNetConfig(
        name=f"mobilenet_v2_{_n}", url=f"[timm]mobilenetv2_{_n}", stem_size=32 if _fix else round_channels(32, _cm),
        architecture=_ARCH_MNV2, channel_multiplier=_cm, depth_multiplier=_dm, fix_first_last=_fix,
        nb_features=1280 if _fix else max(1280, round_channels(1280, _cm)), norm_layer="batch_norm", act_layer="relu6",
        padding="symmetric"))


def _register(cfg):
    def fn():
        return EfficientNet, cfg
    fn.__name__ = fn.__qualname__ = cfg.name
    fn.__module__ = __name__
    fn.__doc__ = f"{cfg.name} (reference tfimm/architectures/efficientnet.py)"
    globals()[cfg.name] = register_model(fn)


for _cfg in _CFGS:
    _register(_cfg)
del _cfg, _CFGS
