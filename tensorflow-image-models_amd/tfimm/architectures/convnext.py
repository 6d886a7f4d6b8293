"""ConvNeXt on the MI355X engine.

Behavioural mirror of reference tfimm/architectures/convnext.py (ConvNeXtConfig :66-135,
ConvNeXtBlock :147-234, ConvNeXtStage :237-300, ConvNeXt :303-445, registrations :448-679).
Lowering of one block (convnext.py:222-232):

  ZeroPadding2D(3) + DepthwiseConv2D(7) + bias  -> strip depthwise kernel (padding folded in)
  LayerNorm(C)                                  -> wave-per-row LayerNorm on the NHWC rows
  fc1 + exact-erf GELU                          -> GEMM with fused activation
  fc2, * gamma, + shortcut                      -> ONE GEMM: LayerScale is folded into fc2's weights and
                                                   bias on the host (gamma * (W h + b) = (W gamma) h + b gamma),
                                                   the residual add rides in the epilogue
Stem = 4x4/4 conv on the pixel-pair view of the input + LayerNorm; downsample = LayerNorm + 2x2/2 conv
(implicit-GEMM gather); head = mean over rows -> LayerNorm -> fc (convnext.py:431-444).
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

from ..models.config import ModelConfig
from ..models.model import Model, WeightSpec
from ..models.registry import register_model
from ..utils.constants import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD

__all__ = ["ConvNeXt", "ConvNeXtConfig"]

_LN_EPS = {"layer_norm": 1e-5, "layer_norm_eps_1e-6": 1e-6}


@dataclass
class ConvNeXtConfig(ModelConfig):
    nb_classes: int = 1000
    in_channels: int = 3
    input_size: Tuple[int, int] = (224, 224)
    patch_size: int = 4
    embed_dim: Tuple = (96, 192, 384, 768)
    nb_blocks: Tuple = (3, 3, 9, 3)
    mlp_ratio: float = 4.0
    conv_mlp_block: bool = False
    # Regularization
    drop_rate: float = 0.0
    drop_path_rate: float = 0.1
    # Other parameters
    norm_layer: str = "layer_norm_eps_1e-6"
    act_layer: str = "gelu"
    init_scale: float = 1e-6
    # Parameters for inference
    crop_pct: float = 0.875
    interpolation: str = "bicubic"
    # Preprocessing
    mean: Tuple[float, float, float] = IMAGENET_DEFAULT_MEAN
    std: Tuple[float, float, float] = IMAGENET_DEFAULT_STD
    # Weight transfer
    first_conv: str = "stem/0"
    classifier: str = "head/fc"


class ConvNeXt(Model):
    cfg_class = ConvNeXtConfig

    def weight_specs(self):
        c = self.cfg
        s = OrderedDict()
        s["stem/0/kernel"] = WeightSpec((c.patch_size, c.patch_size, c.in_channels, c.embed_dim[0]), "conv")
        s["stem/0/bias"] = WeightSpec((c.embed_dim[0],), "bias")
        s["stem/1/gamma"] = WeightSpec((c.embed_dim[0],), "gamma")
        s["stem/1/beta"] = WeightSpec((c.embed_dim[0],), "beta")
        for j, (D, nb) in enumerate(zip(c.embed_dim, c.nb_blocks)):
            if j > 0:
                Dp = c.embed_dim[j - 1]
                s[f"stages/{j}/downsample/0/gamma"] = WeightSpec((Dp,), "gamma")
                s[f"stages/{j}/downsample/0/beta"] = WeightSpec((Dp,), "beta")
                s[f"stages/{j}/downsample/1/kernel"] = WeightSpec((2, 2, Dp, D), "conv")
                s[f"stages/{j}/downsample/1/bias"] = WeightSpec((D,), "bias")
            Hd = int(c.mlp_ratio * D)
            for i in range(nb):
                p = f"stages/{j}/blocks/{i}/"
                s[p + "conv_dw/depthwise_kernel"] = WeightSpec((7, 7, D, 1), "dwconv")
                s[p + "conv_dw/bias"] = WeightSpec((D,), "bias")
                s[p + "norm/gamma"] = WeightSpec((D,), "gamma")
                s[p + "norm/beta"] = WeightSpec((D,), "beta")
                k1 = (1, 1, D, Hd) if c.conv_mlp_block else (D, Hd)       # ConvMLP vs MLP (layers/transformers.py)
                k2 = (1, 1, Hd, D) if c.conv_mlp_block else (Hd, D)
                s[p + "mlp/fc1/kernel"] = WeightSpec(k1, "conv" if c.conv_mlp_block else "dense")
                s[p + "mlp/fc1/bias"] = WeightSpec((Hd,), "bias")
                s[p + "mlp/fc2/kernel"] = WeightSpec(k2, "conv" if c.conv_mlp_block else "dense")
                s[p + "mlp/fc2/bias"] = WeightSpec((D,), "bias")
                s[p + "gamma"] = WeightSpec((D,), "scale", init=str(c.init_scale))
        Dl = c.embed_dim[-1]
        s["head/norm/gamma"] = WeightSpec((Dl,), "gamma")
        s["head/norm/beta"] = WeightSpec((Dl,), "beta")
        if c.nb_classes > 0:
            s["head/fc/kernel"] = WeightSpec((Dl, c.nb_classes), "dense")
            s["head/fc/bias"] = WeightSpec((c.nb_classes,), "bias")
        self.nb_features = Dl
        return s

    @property
    def feature_names(self) -> List[str]:
        names = ["stem"]
        for j, nb in enumerate(self.cfg.nb_blocks):
            if j > 0:
                names.append(f"stage_{j}/downsample")
            names += [f"stage_{j}/block_{i}" for i in range(nb)]
        return names + ["conv_features", "features", "logits"]

    def lower(self, b, H, W, want_features):
        c = self.cfg
        eps = _LN_EPS[c.norm_layer]
        x = b.image_input(H, W, c.in_channels)
        x = b.conv(x, "stem/0/kernel", stride=c.patch_size, padding=0, bias="stem/0/bias", cite="convnext.py:404")
        x = b.layernorm(x, "stem/1", eps, cite="convnext.py:405")
        if want_features:
            b.p.mark_output("stem", x)
        for j, nb in enumerate(c.nb_blocks):
            if j > 0:
                y = b.layernorm(x, f"stages/{j}/downsample/0", eps, cite="convnext.py:292")
                x = b.conv(y, f"stages/{j}/downsample/1/kernel", stride=2, padding=0, bias=f"stages/{j}/downsample/1/bias",
                           cite="convnext.py:293")
                if want_features:
                    b.p.mark_output(f"stage_{j}/downsample", x)
            for i in range(nb):
                p = f"stages/{j}/blocks/{i}/"
                y, _ = b.dwconv(x, p + "conv_dw/depthwise_kernel", stride=1, padding=3, bias=p + "conv_dw/bias",
                                cite="convnext.py:224-225")
                z = b.mlp_fused(y, p + "norm", eps, p + "mlp/fc1", p + "mlp/fc2", act=c.act_layer, residual=x,
                                out_scale=p + "gamma", cite="convnext.py:226-232, transformers.py:208-214")
                if z is None:
                    h = b.ln_dense(y, p + "norm", eps, p + "mlp/fc1/kernel", p + "mlp/fc1/bias", act=c.act_layer,
                                   cite_ln="convnext.py:226", cite="transformers.py:209-210")
                    z = b.dense(h, p + "mlp/fc2/kernel", p + "mlp/fc2/bias", out_scale=p + "gamma", residual=x,
                                cite="transformers.py:212 + convnext.py:228-230")
                x = z
                if want_features:
                    b.p.mark_output(f"stage_{j}/block_{i}", x)
        b.p.mark_output("conv_features", x)
        pooled = b.mean_rows(x, cite="convnext.py:431")
        feat = b.layernorm(pooled, "head/norm", eps, cite="convnext.py:432")
        b.p.mark_output("features", feat)
        if c.nb_classes > 0:
            logits = b.dense(feat, "head/fc/kernel", "head/fc/bias", out_f32=True, cite="convnext.py:436")
        else:
            logits = feat
        b.p.mark_output("logits", logits)

    def forward_features(self, x, training: bool = False, return_features: bool = False):
        """Pre-pooling feature map (convnext.py:383-411), NHWC."""
        if training:
            raise NotImplementedError("This engine implements the inference forward path only (training=False).")
        out = self._run(x, return_features)
        return self._finish(out, "conv_features", return_features)


def _cn(name, embed, blocks, **kw):
    return ConvNeXtConfig(**{**dict(name=name, url="[timm]", embed_dim=embed, nb_blocks=blocks), **kw})


def _register(cfg):
    def fn():
        return ConvNeXt, cfg
    fn.__name__ = fn.__qualname__ = cfg.name
    fn.__module__ = __name__
    fn.__doc__ = f"{cfg.name} (reference tfimm/architectures/convnext.py)"
    globals()[cfg.name] = register_model(fn)


_SIZES = {"tiny": ((96, 192, 384, 768), (3, 3, 9, 3)), "small": ((96, 192, 384, 768), (3, 3, 27, 3)),
          "base": ((128, 256, 512, 1024), (3, 3, 27, 3)), "large": ((192, 384, 768, 1536), (3, 3, 27, 3)),
          "xlarge": ((256, 512, 1024, 2048), (3, 3, 27, 3))}
for _s in ("tiny", "small", "base", "large"):
    _register(_cn(f"convnext_{_s}", *_SIZES[_s]))
for _s in ("tiny", "small", "base", "large", "xlarge"):
    _register(_cn(f"convnext_{_s}_in22ft1k", *_SIZES[_s]))
for _s in ("tiny", "small", "base", "large", "xlarge"):
    _register(_cn(f"convnext_{_s}_384_in22ft1k", *_SIZES[_s], input_size=(384, 384)))
for _s in ("tiny", "small", "base", "large", "xlarge"):
    _register(_cn(f"convnext_{_s}_in22k", *_SIZES[_s], nb_classes=21841))
del _s
