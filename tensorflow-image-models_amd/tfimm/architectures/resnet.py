"""ResNet family on the MI355X engine.

Behavioural mirror of reference tfimm/architectures/resnet.py (ResNetConfig :55-99,
BasicBlock :102-189, Bottleneck :192-292, downsample_avg/_conv :295-330, make_stage
:333-382, ResNet :385-593, registrations :596-1705).  Lowering:

  every Conv2D(+ZeroPadding2D) + BatchNormalization(inference) + ReLU group is ONE
  implicit-GEMM kernel launch with the BN folded into the packed weights and the ReLU in
  the epilogue; the block's shortcut add + final ReLU ride in the epilogue of its last conv
  (resnet.py:266-292); the 7x7 stem reads the padded-RGB image directly (no im2col).

ResNet-D shortcuts (average pooling + 1x1 conv) run as one folded 2x2 stride-2 convolution at even sizes and as
tfimm_hip_avg_pool + 1x1 conv at odd ones; ResNeXt's grouped 3x3 as a dense convolution over the block-diagonal kernel;
ECA as fp32 channel means -> tfimm_hip_eca_gate -> scale; GroupNorm models (norm_layer="group_norm") keep the
normalisation as its own kernel behind each convolution (tfimm_hip_group_norm, with the activation and the shortcut add
fused); BlurPool anti-aliasing is tfimm_hip_blur_pool.
"""
import math
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from ..models.config import ModelConfig
from ..models.model import Model, WeightSpec
from ..models.registry import register_model
from ..utils.constants import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD
from ..utils.etc import make_divisible

__all__ = ["ResNet", "ResNetConfig"]

_BN_EPS = {"batch_norm": 1e-5, "batch_norm_tf": 1e-3}  # layers/factory.py:22-37


@dataclass
class ResNetConfig(ModelConfig):
    nb_classes: int = 1000
    in_channels: int = 3
    input_size: Tuple[int, int] = (224, 224)
    # Residual blocks
    block: str = "basic_block"
    nb_blocks: Tuple = (2, 2, 2, 2)
    nb_channels: Tuple = (64, 128, 256, 512)
    cardinality: int = 1
    base_width: int = 64
    downsample_mode: str = "conv"
    zero_init_last_bn: bool = True
    # Stem
    stem_width: int = 64
    stem_type: str = ""
    replace_stem_pool: bool = False
    # Other params
    block_reduce_first: int = 1
    down_kernel_size: int = 1
    act_layer: str = "relu"
    norm_layer: str = "batch_norm"
    aa_layer: str = ""
    attn_layer: str = ""
    se_ratio: float = 0.0625
    # Regularization
    drop_rate: float = 0.0
    drop_path_rate: float = 0.0
    # Head
    global_pool: str = "avg"
    # Parameters for inference
    test_input_size: Optional[Tuple[int, int]] = None
    pool_size: int = 7
    crop_pct: float = 0.875
    interpolation: str = "bilinear"
    # Preprocessing
    mean: Tuple[float, float, float] = IMAGENET_DEFAULT_MEAN
    std: Tuple[float, float, float] = IMAGENET_DEFAULT_STD
    # Weight transfer
    first_conv: str = "conv1"
    classifier: str = "fc"

    def __post_init__(self):
        if self.test_input_size is None:
            self.test_input_size = self.input_size


#: largest dense block-diagonal expansion of a grouped 3x3 kernel the fallback path accepts (bytes of bf16 per block)
_DENSE_GROUPED_LIMIT = 160 * 2**20
_GN_GROUPS, _GN_EPS = 32, 1e-5      # GroupNormalization defaults (layers/norm.py:128-139, layers/factory.py:55-56)


def _norm_specs(s, prefix, c, norm_layer="batch_norm", zero_init=False):
    """Variables of one norm layer: BatchNormalization has four, GroupNormalization gamma / beta only."""
    z = "zeros" if zero_init else ""
    s[prefix + "/gamma"] = WeightSpec((c,), "gamma", z)
    s[prefix + "/beta"] = WeightSpec((c,), "beta")
    if norm_layer in _BN_EPS:
        s[prefix + "/moving_mean"] = WeightSpec((c,), "mean")
        s[prefix + "/moving_variance"] = WeightSpec((c,), "var", z)


class ResNet(Model):
    cfg_class = ResNetConfig
    keys_to_ignore_on_load = ("blur_kernel",)

    # ---- architecture walk shared by weight_specs() and lower() -----------------------------------
    def _stages(self):
        """Yields (stage idx, block idx, in_ch, nb_channels, out_ch, stride, has_downsample)."""
        c = self.cfg
        expansion = 1 if c.block == "basic_block" else 4
        in_ch = c.stem_width * 2 if c.stem_type in ("deep", "deep_tiered") else 64
        for idx in range(4):
            nb_ch = c.nb_channels[idx]
            out_ch = nb_ch * expansion
            for bidx in range(c.nb_blocks[idx]):
                stride = 1 if idx == 0 or bidx > 0 else 2
                down = bidx == 0 and (stride != 1 or in_ch != out_ch)
                yield idx, bidx, in_ch, nb_ch, out_ch, stride, down
                # NOTE: the reference sets in_channels = nb_channels (not out_channels) after
                # each block (resnet.py:380); only the block-0 test uses it, and for Bottleneck
                # nets nb_channels != out_channels so later stages always get a downsample
                # layer, exactly as here.
                in_ch = nb_ch
            in_ch = nb_ch

    def _block_widths(self, nb_ch):
        c = self.cfg
        if c.block == "basic_block":
            first = nb_ch // c.block_reduce_first
            return first, nb_ch, nb_ch
        width = int(math.floor(nb_ch * (c.base_width / 64)) * c.cardinality)   # resnet.py:213
        return width // c.block_reduce_first, width, nb_ch * 4

    def weight_specs(self):
        c = self.cfg
        s = OrderedDict()

        def _bn_specs(s_, prefix, ch_, zero_init=False):
            _norm_specs(s_, prefix, ch_, c.norm_layer, zero_init)

        if c.stem_type in ("deep", "deep_tiered"):
            ch = (3 * (c.stem_width // 4), c.stem_width) if c.stem_type == "deep_tiered" else (c.stem_width, c.stem_width)
            s["conv1/0/kernel"] = WeightSpec((3, 3, c.in_channels, ch[0]), "conv")
            _bn_specs(s, "conv1/1", ch[0])
            s["conv1/3/kernel"] = WeightSpec((3, 3, ch[0], ch[1]), "conv")
            _bn_specs(s, "conv1/4", ch[1])
            s["conv1/6/kernel"] = WeightSpec((3, 3, ch[1], c.stem_width * 2), "conv")
            stem_out = c.stem_width * 2
        else:
            s["conv1/kernel"] = WeightSpec((7, 7, c.in_channels, 64), "conv")
            stem_out = 64
        _bn_specs(s, "bn1", stem_out)
        if c.replace_stem_pool:
            s["maxpool/0/kernel"] = WeightSpec((3, 3, stem_out, stem_out), "conv")
            _bn_specs(s, "maxpool/1", stem_out)
        prev_out = stem_out
        for idx, bidx, in_ch, nb_ch, out_ch, stride, down in self._stages():
            p = f"layer{idx + 1}/{bidx}"
            first, width, outp = self._block_widths(nb_ch)
            zi = c.zero_init_last_bn
            if c.block == "basic_block":
                s[p + "/conv1/kernel"] = WeightSpec((3, 3, prev_out, first), "conv")
                _bn_specs(s, p + "/bn1", first)
                s[p + "/conv2/kernel"] = WeightSpec((3, 3, first, outp), "conv")
                _bn_specs(s, p + "/bn2", outp, zero_init=zi)
            else:
                s[p + "/conv1/kernel"] = WeightSpec((1, 1, prev_out, first), "conv")
                _bn_specs(s, p + "/bn1", first)
                s[p + "/conv2/kernel"] = WeightSpec((3, 3, first // c.cardinality, width), "conv")
                _bn_specs(s, p + "/bn2", width)
                s[p + "/conv3/kernel"] = WeightSpec((1, 1, width, outp), "conv")
                _bn_specs(s, p + "/bn3", outp, zero_init=zi)
            if c.attn_layer == "se":
                rd = make_divisible(outp * c.se_ratio, 8, round_limit=0.0)     # layers/attention.py:50-52
                s[p + "/se/fc1/kernel"] = WeightSpec((1, 1, outp, rd), "conv")
                s[p + "/se/fc1/bias"] = WeightSpec((rd,), "bias")
                s[p + "/se/fc2/kernel"] = WeightSpec((1, 1, rd, outp), "conv")
                s[p + "/se/fc2/bias"] = WeightSpec((outp,), "bias")
            elif c.attn_layer == "eca":
                t = int(abs(math.log(outp, 2) + 1) / 2)                          # layers/attention.py:104-106
                k = max(t if t % 2 else t + 1, 3)
                s[p + "/se/conv/kernel"] = WeightSpec((k, 1, 1), "conv")
            if down:
                if c.downsample_mode == "avg":
                    s[p + "/downsample/1/kernel"] = WeightSpec((1, 1, prev_out, out_ch), "conv")
                    _bn_specs(s, p + "/downsample/2", out_ch)
                else:
                    k = c.down_kernel_size
                    s[p + "/downsample/0/kernel"] = WeightSpec((k, k, prev_out, out_ch), "conv")
                    _bn_specs(s, p + "/downsample/1", out_ch)
            prev_out = out_ch
        if c.nb_classes > 0:
            s["remove/fc/kernel"] = WeightSpec((prev_out, c.nb_classes), "dense")
            s["remove/fc/bias"] = WeightSpec((c.nb_classes,), "bias")
        self.nb_features = prev_out
        return s

    @property
    def feature_names(self) -> List[str]:
        return ["stem"] + [f"block_{j}" for j in range(sum(self.cfg.nb_blocks))] + ["features", "logits"]

    # ---- lowering ---------------------------------------------------------------------------------------
    def check_supported(self) -> None:
        """Configuration features this engine does not lower (independent of weights and input size)."""
        c = self.cfg
        if c.norm_layer not in _BN_EPS and c.norm_layer != "group_norm":
            raise NotImplementedError(f"norm_layer={c.norm_layer!r} is not built.")
        if c.aa_layer not in ("", "blur_pool"):
            raise NotImplementedError(f"aa_layer={c.aa_layer!r} is not built.")
        if c.attn_layer not in ("", "se", "eca"):
            raise NotImplementedError(f"attn_layer={c.attn_layer!r} is not built yet.")
        if c.global_pool != "avg":
            raise NotImplementedError("only global_pool='avg' is built.")

    def lower(self, b, H, W, want_features):
        c = self.cfg
        self.check_supported()
        gn = c.norm_layer == "group_norm"
        eps = _GN_EPS if gn else _BN_EPS[c.norm_layer]
        act = c.act_layer

        def conv_norm(x, kernel, norm, *, act="", residual=None, act_after=False, cite="", **kw):
            """Conv2D -> norm -> activation (-> + shortcut -> activation).  BatchNorm folds into the convolution's
            weights and everything rides in its epilogue; GroupNorm statistics depend on the convolution's whole
            output, so it runs as its own kernel with the activation / shortcut add fused into it instead."""
            if not gn:
                return b.conv(x, kernel, bn=norm, bn_eps=eps, act=act, residual=residual, act_after_res=act_after,
                              cite=cite, **kw)
            y = b.conv(x, kernel, cite=cite, **kw)
            if residual is not None:        # norm -> + shortcut -> act  (resnet.py:281-290)
                return b.group_norm(y, norm, _GN_GROUPS, eps, residual=residual, act_after=act,
                                    cite="layers/norm.py:37-165")
            return b.group_norm(y, norm, _GN_GROUPS, eps, act=act, cite="layers/norm.py:37-165")

        x = b.image_input(H, W, c.in_channels)
        # ---- stem (resnet.py:466-512, 572-575)
        fused_pool = False
        if c.stem_type in ("deep", "deep_tiered"):
            x = conv_norm(x, "conv1/0/kernel", "conv1/1", stride=2, padding=1, act=act, cite="resnet.py:473-481")
            x = conv_norm(x, "conv1/3/kernel", "conv1/4", padding="same", act=act, cite="resnet.py:482-490")
            x = conv_norm(x, "conv1/6/kernel", "bn1", padding="same", act=act, cite="resnet.py:491-500,513-514")
        else:
            # plain stem: convolution and the pooling behind it go to the builder together (one kernel when it can)
            fused_pool = not c.replace_stem_pool and not c.aa_layer and not gn
            x = conv_norm(x, "conv1/kernel", "bn1", stride=2, padding=3, act=act,
                          cite="resnet.py:505-514,538-540", **({"then_maxpool": (3, 2, 1)} if fused_pool else {}))
        # ---- stem pooling (resnet.py:517-540)
        if c.replace_stem_pool:
            x = conv_norm(x, "maxpool/0/kernel", "maxpool/1", stride=2, padding=1, act=act, cite="resnet.py:520-530")
        elif c.aa_layer:
            x = b.maxpool(x, 3, 1, 1, cite="resnet.py:533-534")
            x = b.blur_pool(x, 2, cite="resnet.py:535 + layers/blurpool.py:52-60")
        elif not fused_pool:
            x = b.maxpool(x, 3, 2, 1, cite="resnet.py:538-540")
        if want_features:
            b.p.mark_output("stem", x)
        # ---- residual stages
        j = 0
        for idx, bidx, in_ch, nb_ch, out_ch, stride, down in self._stages():
            p = f"layer{idx + 1}/{bidx}"
            shortcut = x
            ds_spec = None
            fold_spec = None            # (x, kernel, bn, stride, emit): the shortcut convolution as a second operand of conv3
            if down:
                if c.downsample_mode == "avg":
                    # resnet.py:295-312 (ResNet-D): AveragePooling2D(2, stride, "same") -> 1x1 conv -> norm.  At even
                    # sizes "same" adds no padding and every window holds 4 valid elements, so pooling + 1x1 conv IS
                    # a 2x2 stride-2 convolution with kernel / 4 on every tap (exact in bf16: a power of two) -- one
                    # GEMM launch, no pooled intermediate.  At odd sizes the last row / column of windows is clipped
                    # and averages over 2 or 1 elements: explicit pooling kernel, then the 1x1 convolution.
                    kname = p + "/downsample/1/kernel"
                    if stride == 1:
                        shortcut = conv_norm(x, kname, p + "/downsample/2", cite="resnet.py:295-312")
                    elif stride == 2 and x.H % 2 == 0 and x.W % 2 == 0:
                        def emit_shortcut(x=x, p=p, kname=kname):
                            folded = b.define(kname + ":avgpool2x2", np.tile(b.wget(kname) * 0.25, (2, 2, 1, 1)))
                            return conv_norm(x, folded, p + "/downsample/2", stride=2, padding="valid", flops_k=x.C,
                                             cite="resnet.py:295-312")
                        if not gn and b.can_fold_shortcut(x, 2, out_ch):
                            # ... or not even that: the four taps as a 2 x 2 window of the block's last convolution's second operand
                            shortcut = None
                            fold_spec = (x, kname, p + "/downsample/2", 2, emit_shortcut, 2)
                        else:
                            shortcut = emit_shortcut()
                    else:
                        pooled = b.avg_pool(x, 2, stride, cite="resnet.py:299-301")
                        shortcut = conv_norm(pooled, kname, p + "/downsample/2", cite="resnet.py:304-312")
                else:
                    pd = (stride + c.down_kernel_size) // 2 - 1                       # resnet.py:319

                    def emit_shortcut(x=x, p=p, stride=stride, pd=pd):
                        return conv_norm(x, p + "/downsample/0/kernel", p + "/downsample/1", stride=stride, padding=pd,
                                         cite="resnet.py:315-330")
                    # a 1x1 / stride 1 shortcut convolution of a 64-channel input (first block of stage 1) can be multiplied
                    # inside the fused bottleneck tail: decided below, emitted here otherwise
                    if (c.block == "bottleneck" and stride == 1 and c.down_kernel_size == 1 and x.C == 64 and not gn
                            and os.environ.get("TFIMM_NO_CHAIN_SHORTCUT", "0") != "1"):
                        shortcut = None
                        ds_spec = (x, p + "/downsample/0/kernel", p + "/downsample/1", emit_shortcut)
                    elif (c.down_kernel_size == 1 and not gn and b.can_fold_shortcut(x, stride, out_ch)):
                        # any other 1x1 shortcut convolution (the strided first blocks of stages 2 - 4): its input channels become
                        # further k-tiles of the block's last convolution -- conv3 of a bottleneck, the 3x3 conv2 of a basic block
                        # (tfimm_gemm_desc::a2) -- no shortcut tensor, no launch
                        shortcut = None
                        fold_spec = (x, p + "/downsample/0/kernel", p + "/downsample/1", stride, emit_shortcut)
                    else:
                        shortcut = emit_shortcut()
            se = c.attn_layer == "se"
            gated = c.attn_layer in ("se", "eca")       # the gate sits between the last norm and the shortcut add
            use_aa = bool(c.aa_layer) and stride == 2                                  # resnet.py:127,218
            if ds_spec is not None and (gated or use_aa or c.cardinality != 1 or act != "relu"):
                shortcut, ds_spec = ds_spec[3](), None       # the fused tail will not be used: emit the shortcut now
            if fold_spec is not None and gated:              # the gate multiplies conv3's output only: the add stays separate
                shortcut, fold_spec = fold_spec[4](), None
            last = dict(residual=None if gated else shortcut, act="" if gated else act, act_after=not gated)
            cstride = 1 if use_aa else stride
            if c.block == "basic_block":
                y = conv_norm(x, p + "/conv1/kernel", p + "/bn1", stride=cstride, padding=1, act=act,
                              cite="resnet.py:168-172")
                if use_aa:
                    y = b.blur_pool(y, stride, cite="resnet.py:173-174")
                if fold_spec is not None and not b.can_fold_shortcut(fold_spec[0], fold_spec[3], out_ch, y.C):
                    shortcut, fold_spec = fold_spec[4](), None
                    last = dict(residual=shortcut, act=act, act_after=True)
                if fold_spec is not None:
                    y = conv_norm(y, p + "/conv2/kernel", p + "/bn2", padding=1, act=act, fold_shortcut=fold_spec[:4] + fold_spec[5:],
                                  cite="resnet.py:176-186 + 315-330")
                else:
                    y = conv_norm(y, p + "/conv2/kernel", p + "/bn2", padding=1, cite="resnet.py:176-186", **last)
            else:
                y = conv_norm(x, p + "/conv1/kernel", p + "/bn1", act=act, cite="resnet.py:269-271")
                k2 = p + "/conv2/kernel"
                kw2 = {}
                grouped = None
                if c.cardinality > 1 and not gn:
                    # ResNeXt (resnet.py:229-236, Conv2D(groups=cardinality)): 32-channel super-groups on the MFMA
                    # (tfimm_hip_grouped_conv3x3) when a group has at most 32 channels
                    grouped = b.grouped_conv3x3(y, k2, c.cardinality, stride=cstride, bn=p + "/bn2", bn_eps=eps, act=act,
                                                cite="resnet.py:229-241,273-276")
                if c.cardinality > 1 and grouped is None and not gn:
                    # groups of 64+ channels (ResNeXt-101 32x16d / 32d / 48d, the late stages of 32x8d): one implicit-GEMM
                    # launch per group on its channel slice
                    grouped = b.grouped_conv_split(y, k2, c.cardinality, stride=cstride, padding=1, bn=p + "/bn2", bn_eps=eps,
                                                   act=act, cite="resnet.py:229-241,273-276")
                if c.cardinality > 1 and grouped is None:
                    # what is left (group widths that are not multiples of 8, GroupNorm variants): a dense convolution over
                    # the block-diagonal expansion of the kernel -- exact (the extra products are x * 0), at cardinality x the
                    # multiply-accumulates and weight bytes
                    kg = b.wget(k2)
                    dense_bytes = 9 * kg.shape[3] * kg.shape[3] * 2
                    if dense_bytes > _DENSE_GROUPED_LIMIT:
                        raise NotImplementedError(
                            f"{c.name}: grouped 3x3 with {kg.shape[2]} channels per group would expand to a "
                            f"{dense_bytes / 2**20:.0f} MiB dense kernel per block (limit {_DENSE_GROUPED_LIMIT / 2**20:.0f} MiB)")
                    k2 = b.define(k2 + ":dense", _expand_grouped_kernel(kg, c.cardinality))
                    kw2["flops_k"] = 9 * y.C // c.cardinality
                fused = None
                if not gn and not gated and not use_aa and c.cardinality == 1 and act == "relu" and fold_spec is None:
                    # conv2 + bn2 + act2 + conv3 + bn3 + shortcut add + act3 as one launch: the `width`-channel
                    # intermediate stays in LDS (stages 1 and 2; None for shapes that kernel is not built for)
                    fused = b.conv_chain(y, k2, p + "/bn2", p + "/conv3/kernel", p + "/bn3", stride=cstride, padding=1,
                                         bn_eps=eps, act1=act, act2=act, residual=shortcut,
                                         shortcut_conv=None if ds_spec is None else ds_spec[:3], cite="resnet.py:273-290")
                if ds_spec is not None and fused is None:       # shape outside the fused kernel: the shortcut as its own launch
                    shortcut = ds_spec[3]()
                    last = dict(residual=shortcut, act=act, act_after=True)
                if fused is not None:
                    y = fused
                else:
                    y = grouped if grouped is not None else conv_norm(y, k2, p + "/bn2", stride=cstride, padding=1, act=act,
                                                                      cite="resnet.py:273-276", **kw2)
                    if use_aa:
                        y = b.blur_pool(y, stride, cite="resnet.py:277-278")
                    if fold_spec is not None and not b.can_fold_shortcut(fold_spec[0], fold_spec[3], out_ch, y.C):
                        shortcut, fold_spec = fold_spec[4](), None          # conv3's own input is not 16-byte aligned: the plain form
                        last = dict(residual=shortcut, act=act, act_after=True)
                    if fold_spec is not None:
                        y = conv_norm(y, p + "/conv3/kernel", p + "/bn3", act=act, fold_shortcut=fold_spec[:4] + fold_spec[5:],
                                      cite="resnet.py:280-290 + 315-330")
                    else:
                        y = conv_norm(y, p + "/conv3/kernel", p + "/bn3", cite="resnet.py:280-290", **last)
            if c.attn_layer == "eca":
                # EcaModule (layers/attention.py:105-130): fp32 channel means -> Conv1D over the channel axis -> sigmoid
                m = b.mean_rows(y, out_f32=True, cite="layers/attention.py:122")
                g = b.eca_gate(m, p + "/se/conv/kernel", cite="layers/attention.py:123-126")
                y = b.scale_channels(y, g, residual=shortcut, relu_after=True,
                                     cite="layers/attention.py:129 + resnet.py:289-290")
            if se:
                m = b.mean_rows(y, out_f32=True, cite="layers/attention.py:67")
                g = b.se_gate(m, 1, p + "/se/fc1/kernel", p + "/se/fc1/bias", p + "/se/fc2/kernel", p + "/se/fc2/bias",
                              act="relu", cite="layers/attention.py:68-72")
                y = b.scale_channels(y, g, residual=shortcut, relu_after=True, cite="layers/attention.py:73 + resnet.py:289-290")
            x = y
            if want_features:
                b.p.mark_output(f"block_{j}", x)
            j += 1
        b.p.mark_output("features", x)
        # ---- head (layers/classifier.py:65-74)
        pooled = b.mean_rows(x, cite="layers/classifier.py:66-67")
        if c.nb_classes > 0:
            logits = b.dense(pooled, "remove/fc/kernel", "remove/fc/bias", out_f32=True, cite="layers/classifier.py:70-71")
        else:
            logits = pooled
        b.p.mark_output("logits", logits)


# ---------------------------------------------------------------------------------------
# registrations (reference resnet.py:596-1705)
# ---------------------------------------------------------------------------------------
def _expand_grouped_kernel(k: np.ndarray, groups: int) -> np.ndarray:
    """(kh, kw, Cin / groups, Cout) grouped kernel -> (kh, kw, Cin, Cout) with zeros outside the diagonal blocks
    (tf.keras Conv2D(groups=g): output channel o reads input channels of group o // (Cout / g))."""
    kh, kw, cg, cout = k.shape
    og = cout // groups
    dense = np.zeros((kh, kw, cg * groups, cout), dtype=k.dtype)
    for g in range(groups):
        dense[:, :, g * cg:(g + 1) * cg, g * og:(g + 1) * og] = k[:, :, :, g * og:(g + 1) * og]
    return dense


_D = dict(stem_width=32, stem_type="deep", downsample_mode="avg", first_conv="conv1/0")
_T = dict(stem_width=32, stem_type="deep_tiered", downsample_mode="avg", first_conv="conv1/0")
_BC = dict(interpolation="bicubic")
_L = {18: ("basic_block", (2, 2, 2, 2)), 26: ("bottleneck", (2, 2, 2, 2)), 34: ("basic_block", (3, 4, 6, 3)),
      50: ("bottleneck", (3, 4, 6, 3)), 101: ("bottleneck", (3, 4, 23, 3)), 152: ("bottleneck", (3, 8, 36, 3)),
      200: ("bottleneck", (3, 24, 36, 3))}


def _rn(name, depth, **kw):
    block, blocks = _L[depth]
    return ResNetConfig(**{**dict(name=name, url="[timm]", block=block, nb_blocks=blocks), **kw})


def _x(card, width):
    return dict(cardinality=card, base_width=width)


def _rs(name, blocks, res, test, pool, crop):
    return ResNetConfig(name=name, url="[timm]", block="bottleneck", nb_blocks=blocks, input_size=(res, res),
                        test_input_size=(test, test), pool_size=pool, crop_pct=crop, attn_layer="se", se_ratio=0.25,
                        replace_stem_pool=True, **_D, **_BC)


def _register(cfg):
    def fn():
        return ResNet, cfg
    fn.__name__ = fn.__qualname__ = cfg.name
    fn.__module__ = __name__
    fn.__doc__ = f"{cfg.name} (reference tfimm/architectures/resnet.py)"
    globals()[cfg.name] = register_model(fn)


_BIG = dict(input_size=(256, 256), test_input_size=(320, 320), pool_size=8, crop_pct=1.0)
for _cfg in [
    _rn("resnet18", 18), _rn("resnet18d", 18, **_D, **_BC), _rn("resnet34", 34), _rn("resnet34d", 34, **_D, **_BC),
    _rn("resnet26", 26, **_BC), _rn("resnet26d", 26, **_D, **_BC),
    _rn("resnet26t", 26, input_size=(256, 256), pool_size=8, crop_pct=0.94, **_T, **_BC),
    _rn("resnet50", 50, crop_pct=0.95, **_BC), _rn("resnet50d", 50, **_D, **_BC),
    _rn("resnet101", 101, crop_pct=0.95, **_BC), _rn("resnet101d", 101, **_D, **_BC, **_BIG),
    _rn("resnet152", 152, crop_pct=0.95, **_BC), _rn("resnet152d", 152, **_D, **_BC, **_BIG),
    _rn("resnet200d", 200, **_D, **_BC, **_BIG),
    _rn("tv_resnet34", 34), _rn("tv_resnet50", 50), _rn("tv_resnet101", 101), _rn("tv_resnet152", 152),
    _rn("wide_resnet50_2", 50, base_width=128, **_BC), _rn("wide_resnet101_2", 101, base_width=128),
    _rn("resnet50_gn", 50, crop_pct=0.94, norm_layer="group_norm", **_BC),
    _rn("resnext50_32x4d", 50, crop_pct=0.95, **_x(32, 4), **_BC), _rn("resnext50d_32x4d", 50, **_x(32, 4), **_D, **_BC),
    _rn("resnext101_32x8d", 101, **_x(32, 8)), _rn("tv_resnext50_32x4d", 50, **_x(32, 4)),
    _rn("ig_resnext101_32x8d", 101, **_x(32, 8)), _rn("ig_resnext101_32x16d", 101, **_x(32, 16)),
    _rn("ig_resnext101_32x32d", 101, **_x(32, 32)), _rn("ig_resnext101_32x48d", 101, **_x(32, 48)),
    _rn("ssl_resnet18", 18), _rn("ssl_resnet50", 50), _rn("ssl_resnext50_32x4d", 50, **_x(32, 4)),
    _rn("ssl_resnext101_32x4d", 101, **_x(32, 4)), _rn("ssl_resnext101_32x8d", 101, **_x(32, 8)),
    _rn("ssl_resnext101_32x16d", 101, **_x(32, 16)),
    _rn("swsl_resnet18", 18), _rn("swsl_resnet50", 50), _rn("swsl_resnext50_32x4d", 50, **_x(32, 4)),
    _rn("swsl_resnext101_32x4d", 101, **_x(32, 4)), _rn("swsl_resnext101_32x8d", 101, **_x(32, 8)),
    _rn("swsl_resnext101_32x16d", 101, **_x(32, 16)),
    _rn("ecaresnet26t", 26, attn_layer="eca", input_size=(256, 256), test_input_size=(320, 320), pool_size=8,
        crop_pct=0.95, **_T, **_BC),
    _rn("ecaresnetlight", 50, nb_blocks=(1, 1, 11, 3), attn_layer="eca", stem_width=32, downsample_mode="avg", **_BC),
    _rn("ecaresnet50d", 50, attn_layer="eca", **_D, **_BC),
    _rn("ecaresnet50t", 50, attn_layer="eca", test_input_size=(320, 320), pool_size=8, crop_pct=0.95, **_T, **_BC),
    _rn("ecaresnet101d", 101, attn_layer="eca", **_D, **_BC),
    _rn("ecaresnet269d", 50, nb_blocks=(3, 30, 48, 8), attn_layer="eca", input_size=(320, 320),
        test_input_size=(352, 352), pool_size=10, crop_pct=1.0, **_D, **_BC),
    _rn("resnetblur50", 50, aa_layer="blur_pool", **_BC),
    _rn("seresnet50", 50, attn_layer="se", **_BC), _rn("seresnet152d", 152, attn_layer="se", **_D, **_BC, **_BIG),
    _rn("seresnext26d_32x4d", 26, attn_layer="se", **_x(32, 4), **_D, **_BC),
    _rn("seresnext26t_32x4d", 26, attn_layer="se", **_x(32, 4), **_T, **_BC),
    _rn("seresnext50_32x4d", 50, attn_layer="se", **_x(32, 4), **_BC),
    _rs("resnetrs50", (3, 4, 6, 3), 160, 224, 5, 0.91), _rs("resnetrs101", (3, 4, 23, 3), 192, 288, 6, 0.94),
    _rs("resnetrs152", (3, 8, 36, 3), 256, 320, 8, 1.0), _rs("resnetrs200", (3, 24, 36, 3), 256, 320, 8, 1.0),
    _rs("resnetrs270", (4, 29, 53, 4), 256, 352, 8, 1.0), _rs("resnetrs350", (4, 36, 72, 4), 288, 384, 9, 1.0),
    _rs("resnetrs420", (4, 44, 87, 4), 320, 416, 10, 1.0),
]:
    _register(_cfg)
del _cfg
