"""Vision Transformer / DeiT on the MI355X engine.

Behavioural mirror of reference tfimm/architectures/vit.py (ViTConfig :36-119,
ViTMultiHeadAttention :122-171, ViTBlock :174-235, ViT :298-478, registrations :481-1163)
for ``patch_layer="patch_embeddings"``.  The forward pass is lowered to:

  patch-embed conv (k = s = patch) as an im2col-free implicit GEMM that writes straight into
  the token buffer rows 1.. and adds pos_embed in its epilogue          (transformers.py:164-170, vit.py:427-434)
  class/dist token rows = token + pos_embed, broadcast                   (vit.py:427-434)
  per block:  LN -> QKV GEMM -> fused attention -> proj GEMM(+residual)
              LN -> fc1 GEMM(+erf GELU) -> fc2 GEMM(+residual)            (vit.py:219-235, transformers.py:208-214)
  final LN (only on the token rows that feed the head), head GEMM(s)     (vit.py:452-476)
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import numpy as np

from ..layers.transformers import interpolate_pos_embeddings
from ..models.config import ModelConfig
from ..models.model import Model, WeightSpec
from ..models.registry import register_model
from ..utils.constants import (
    IMAGENET_DEFAULT_MEAN,
    IMAGENET_DEFAULT_STD,
    IMAGENET_INCEPTION_MEAN,
    IMAGENET_INCEPTION_STD,
)

__all__ = ["ViT", "ViTConfig"]

_LN_EPS = {"layer_norm": 1e-5, "layer_norm_eps_1e-6": 1e-6}  # layers/factory.py:42-50


@dataclass
class ViTConfig(ModelConfig):
    nb_classes: int = 1000
    in_channels: int = 3
    input_size: Tuple[int, int] = (224, 224)
    patch_layer: str = "patch_embeddings"
    patch_nb_blocks: tuple = ()
    patch_size: int = 16
    embed_dim: int = 768
    nb_blocks: int = 12
    nb_heads: int = 12
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    representation_size: Optional[int] = None
    distilled: bool = False
    # Regularization (identity at inference)
    drop_rate: float = 0.0
    attn_drop_rate: float = 0.0
    drop_path_rate: float = 0.0
    # Other parameters
    norm_layer: str = "layer_norm_eps_1e-6"
    act_layer: str = "gelu"
    # Parameters for inference
    interpolate_input: bool = False
    crop_pct: float = 0.875
    interpolation: str = "bicubic"
    mean: Tuple[float, float, float] = IMAGENET_INCEPTION_MEAN
    std: Tuple[float, float, float] = IMAGENET_INCEPTION_STD
    first_conv: str = "patch_embed/proj"
    classifier: Union[str, Tuple[str, str]] = "head"

    @property
    def nb_tokens(self) -> int:
        return 2 if self.distilled else 1

    @property
    def grid_size(self) -> Tuple[int, int]:
        return (self.input_size[0] // self.patch_size, self.input_size[1] // self.patch_size)

    @property
    def nb_patches(self) -> int:
        return self.grid_size[0] * self.grid_size[1]


class ViT(Model):
    cfg_class = ViTConfig

    def __init__(self, cfg: ViTConfig, *args, **kwargs):
        if cfg.patch_layer != "patch_embeddings":
            raise NotImplementedError(
                "hybrid_embeddings (ResNetV2 backbone) is outside this engine's scope (SURVEY.md §2.1 #10).")
        if cfg.representation_size and cfg.distilled:
            raise ValueError("Cannot combine distillation token and a representation layer.")
        if cfg.norm_layer not in _LN_EPS:
            raise ValueError(f"Unknown normalization layer: {cfg.norm_layer}")
        self.nb_features = cfg.representation_size or cfg.embed_dim
        super().__init__(cfg, *args, **kwargs)

    # -- variables (vit.py:309-400) --------------------------------------------
    def weight_specs(self):
        c = self.cfg
        D, Hd = c.embed_dim, int(c.embed_dim * c.mlp_ratio)
        s = OrderedDict()
        s["patch_embed/proj/kernel"] = WeightSpec((c.patch_size, c.patch_size, c.in_channels, D), "conv")
        s["patch_embed/proj/bias"] = WeightSpec((D,), "bias")
        s["cls_token"] = WeightSpec((1, 1, D), "token")
        if c.distilled:
            s["dist_token"] = WeightSpec((1, 1, D), "token")
        s["pos_embed"] = WeightSpec((1, c.nb_patches + c.nb_tokens, D), "pos")
        for j in range(c.nb_blocks):
            p = f"blocks/{j}/"
            s[p + "norm1/gamma"] = WeightSpec((D,), "gamma")
            s[p + "norm1/beta"] = WeightSpec((D,), "beta")
            s[p + "attn/qkv/kernel"] = WeightSpec((D, 3 * D), "dense")
            if c.qkv_bias:
                s[p + "attn/qkv/bias"] = WeightSpec((3 * D,), "bias")
            s[p + "attn/proj/kernel"] = WeightSpec((D, D), "dense")
            s[p + "attn/proj/bias"] = WeightSpec((D,), "bias")
            s[p + "norm2/gamma"] = WeightSpec((D,), "gamma")
            s[p + "norm2/beta"] = WeightSpec((D,), "beta")
            s[p + "mlp/fc1/kernel"] = WeightSpec((D, Hd), "dense")
            s[p + "mlp/fc1/bias"] = WeightSpec((Hd,), "bias")
            s[p + "mlp/fc2/kernel"] = WeightSpec((Hd, D), "dense")
            s[p + "mlp/fc2/bias"] = WeightSpec((D,), "bias")
        s["norm/gamma"] = WeightSpec((D,), "gamma")
        s["norm/beta"] = WeightSpec((D,), "beta")
        feat = D
        if c.representation_size:
            s["pre_logits/fc/kernel"] = WeightSpec((D, c.representation_size), "dense")
            s["pre_logits/fc/bias"] = WeightSpec((c.representation_size,), "bias")
            feat = c.representation_size
        if c.nb_classes > 0:
            s["head/kernel"] = WeightSpec((feat, c.nb_classes), "dense")
            s["head/bias"] = WeightSpec((c.nb_classes,), "bias")
            if c.distilled:
                s["head_dist/kernel"] = WeightSpec((feat, c.nb_classes), "dense")
                s["head_dist/bias"] = WeightSpec((c.nb_classes,), "bias")
        return s

    @property
    def transform_weights(self):
        """Per-weight adaptations ``transfer_weights`` applies when source and target configs differ
        (vit.py:117-119, 414-420): position embeddings are resized to the target's patch grid."""
        return {"pos_embed": ViT.transform_pos_embed}

    def transform_pos_embed(self, src_weights, target_cfg):
        return interpolate_pos_embeddings(src_weights, self.cfg.grid_size, target_cfg.grid_size, self.cfg.nb_tokens)

    @property
    def feature_names(self) -> List[str]:
        names = ["patch_embedding"]
        for j in range(self.cfg.nb_blocks):
            names += [f"block_{j}/attn", f"block_{j}"]        # vit.py:447-451
        return names + ["features_all", "features", "logits"]

    # -- lowering ------------------------------------------------------------------------------------
    def lower(self, b, H, W, want_features):
        c = self.cfg
        grid = (H // c.patch_size, W // c.patch_size)
        pos = b.wget("pos_embed")
        if (H, W) != tuple(c.input_size):
            if not c.interpolate_input:
                raise ValueError(
                    f"{c.name} was built for {c.input_size} inputs; got {(H, W)}. Create the model with "
                    "interpolate_input=True to resize the position embeddings (vit.py:433-442).")
            # bicubic resize of the patch-grid embeddings, once per input size, on the host (vit.py:436-441)
            pos = interpolate_pos_embeddings(pos, c.grid_size, grid, c.nb_tokens)
        eps = _LN_EPS[c.norm_layer]
        D, nh, nt, npatch = c.embed_dim, c.nb_heads, c.nb_tokens, grid[0] * grid[1]
        N = npatch + nt
        pos = pos[0]                                               # (N, D)
        x = b.image_input(H, W, c.in_channels)
        # patch conv -> token rows [nt, N), + pos_embed[nt:] in the epilogue
        pos_const = b.act_const(pos[nt:], "pos_embed[patches]")
        x = b.conv(x, "patch_embed/proj/kernel", stride=c.patch_size, padding=0, bias="patch_embed/proj/bias",
                   flatten=True, remap=(npatch, N, nt), res_const=pos_const, res_mod=npatch,
                   cite="layers/transformers.py:164-170 + vit.py:427-434", name="tokens")
        toks = [b.wget("cls_token")[0, 0]]
        if c.distilled:
            toks.append(b.wget("dist_token")[0, 0])
        b.token_rows(x, np.stack(toks, 0) + pos[:nt], cite="vit.py:427-434")
        if want_features:
            b.p.mark_output("patch_embedding", x)
        scale = (D // nh) ** -0.5
        for j in range(c.nb_blocks):
            p = f"blocks/{j}/"
            # norm1 / norm2 only feed one Dense layer each: folded into it (gamma into the weights, per-row statistics
            # applied in the GEMM epilogue), so the normalised tensor is never written
            qkv = b.ln_dense(x, p + "norm1", eps, p + "attn/qkv/kernel", p + "attn/qkv/bias" if c.qkv_bias else None,
                             cite_ln="vit.py:222", cite="vit.py:155")
            a = b.attention(qkv, nh, scale, cite="vit.py:156-167", name=p + "attn")
            if want_features:
                # features["block_j/attn"]: the softmax map the fused kernel never writes (vit.py:160-163, 447-450)
                b.p.mark_output(f"block_{j}/attn", b.attention_probs(qkv, nh, scale, cite="vit.py:160-163"))
            x = b.dense(a, p + "attn/proj/kernel", p + "attn/proj/bias", residual=x, cite="vit.py:169,228")
            hdn = b.ln_dense(x, p + "norm2", eps, p + "mlp/fc1/kernel", p + "mlp/fc1/bias", act=c.act_layer,
                             cite_ln="vit.py:231", cite="transformers.py:209-210")
            x = b.dense(hdn, p + "mlp/fc2/kernel", p + "mlp/fc2/bias", residual=x, cite="transformers.py:212, vit.py:234")
            if want_features:
                b.p.mark_output(f"block_{j}", x)
        if want_features:
            allf = b.layernorm(x, "norm", eps, cite="vit.py:452")
            b.p.mark_output("features_all", allf)
        # features: LN only of the rows the head consumes
        if nt == 1:
            features = b.layernorm(x, "norm", eps, row_select=(0, 1), cite="vit.py:452,462", name="norm[cls]")
            if c.representation_size:
                features = b.dense(features, "pre_logits/fc/kernel", "pre_logits/fc/bias", act="tanh",
                                   cite="vit.py:351-359,460")
        else:
            # x[:, :2] after the final LN, kept as one (B, 2*D) tensor (vit.py:458)
            features = b.empty(1, 2 * D, name="features")
            for t in range(2):
                b.layernorm(x, "norm", eps, row_select=(t, 1), out=features, out_col=t * D,
                            cite="vit.py:452,458", name=f"norm[tok{t}]")
        b.p.mark_output("features", features)
        if c.nb_classes > 0:
            if nt == 1:
                logits = b.dense(features, "head/kernel", "head/bias", out_f32=True, cite="vit.py:471-472")
            else:
                logits = b.empty(1, 2 * c.nb_classes, dtype="f32", name="logits")
                b.dense(features, "head/kernel", "head/bias", out_f32=True, in_cols=(0, D), out=logits,
                        out_col=0, cite="vit.py:474")
                b.dense(features, "head_dist/kernel", "head_dist/bias", out_f32=True, in_cols=(D, D),
                        out=logits, out_col=c.nb_classes, cite="vit.py:475-476")
            b.p.mark_output("logits", logits)
        else:
            b.p.mark_output("logits", features)

    def _shape_output(self, name, v):
        c = self.cfg
        if c.distilled and name in ("features", "logits"):
            return v.reshape(v.shape[0], 2, -1)
        if name in ("features", "logits"):
            return v.reshape(v.shape[0], -1)
        if name.endswith("/attn"):
            return v.reshape(v.shape[0], c.nb_heads, v.shape[-1], v.shape[-1])      # (B, H, N, N)
        return v




# ---------------------------------------------------------------------------------------
# registrations (reference vit.py:481-1163); table rows: name -> config overrides
# ---------------------------------------------------------------------------------------
_SIZES = {  # embed_dim, nb_blocks, nb_heads
    "tiny": (192, 12, 3), "small": (384, 12, 6), "base": (768, 12, 12),
    "large": (1024, 24, 16), "huge": (1280, 32, 16),
}


def _vit(name, size, patch, res=224, **kw):
    d, nb, nh = _SIZES[size]
    cfg = dict(name=name, url="[timm]", patch_size=patch, embed_dim=d, nb_blocks=nb, nb_heads=nh)
    if res != 224:
        cfg.update(input_size=(res, res), crop_pct=1.0)
    cfg.update(kw)
    return ViTConfig(**cfg)


def _register(cfg):
    def fn():
        return ViT, cfg
    fn.__name__ = cfg.name
    fn.__qualname__ = cfg.name
    fn.__module__ = __name__
    fn.__doc__ = f"{cfg.name} (reference tfimm/architectures/vit.py)"
    globals()[cfg.name] = register_model(fn)


_IN21K = dict(nb_classes=21843)
_DEIT = dict(mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD)
_DIST = dict(distilled=True, classifier=("head", "head_dist"), **_DEIT)
_MIIL = dict(mean=(0.0, 0.0, 0.0), std=(0.0, 0.0, 0.0), interpolation="bilinear", qkv_bias=False)

for _cfg in [
    _vit("vit_tiny_patch16_224", "tiny", 16),
    _vit("vit_tiny_patch16_384", "tiny", 16, 384),
    _vit("vit_small_patch32_224", "small", 32),
    _vit("vit_small_patch32_384", "small", 32, 384),
    _vit("vit_small_patch16_224", "small", 16),
    _vit("vit_small_patch16_384", "small", 16, 384),
    _vit("vit_base_patch32_224", "base", 32),
    _vit("vit_base_patch32_384", "base", 32, 384),
    _vit("vit_base_patch16_224", "base", 16),
    _vit("vit_base_patch16_384", "base", 16, 384),
    _vit("vit_base_patch8_224", "base", 8),
    _vit("vit_large_patch32_224", "large", 32),
    _vit("vit_large_patch32_384", "large", 32, 384),
    _vit("vit_large_patch16_224", "large", 16),
    _vit("vit_large_patch16_384", "large", 16, 384),
    _vit("vit_base_patch32_sam_224", "base", 32),
    _vit("vit_base_patch16_sam_224", "base", 16),
    _vit("vit_tiny_patch16_224_in21k", "tiny", 16, **_IN21K),
    _vit("vit_small_patch32_224_in21k", "small", 32, **_IN21K),
    _vit("vit_small_patch16_224_in21k", "small", 16, **_IN21K),
    _vit("vit_base_patch32_224_in21k", "base", 32, **_IN21K),
    _vit("vit_base_patch16_224_in21k", "base", 16, **_IN21K),
    _vit("vit_base_patch8_224_in21k", "base", 8, **_IN21K),
    _vit("vit_large_patch32_224_in21k", "large", 32, representation_size=1024, **_IN21K),
    _vit("vit_large_patch16_224_in21k", "large", 16, **_IN21K),
    _vit("vit_huge_patch14_224_in21k", "huge", 14, representation_size=1280, **_IN21K),
    _vit("deit_tiny_patch16_224", "tiny", 16, **_DEIT),
    _vit("deit_small_patch16_224", "small", 16, **_DEIT),
    _vit("deit_base_patch16_224", "base", 16, **_DEIT),
    _vit("deit_base_patch16_384", "base", 16, 384, **_DEIT),
    _vit("deit_tiny_distilled_patch16_224", "tiny", 16, **_DIST),
    _vit("deit_small_distilled_patch16_224", "small", 16, **_DIST),
    _vit("deit_base_distilled_patch16_224", "base", 16, **_DIST),
    _vit("deit_base_distilled_patch16_384", "base", 16, 384, **_DIST),
    _vit("vit_base_patch16_224_miil_in21k", "base", 16, nb_classes=11221, **_MIIL),
    _vit("vit_base_patch16_224_miil", "base", 16, **_MIIL),
]:
    _register(_cfg)
del _cfg
