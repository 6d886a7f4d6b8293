"""CaiT (Class-Attention in Image Transformers) on the MI355X engine.

Behavioural mirror of reference tfimm/architectures/cait.py (CaiTConfig :33-93, ClassAttention
:95-146, LayerScaleBlockClassAttention :149-202, TalkingHeadAttention :205-262, LayerScaleBlock
:265-326, CaiT :329-445, registrations :448-607).  Lowering:

  patch-embed conv (k = s = 16) as implicit GEMM, + pos_embed in its epilogue      (cait.py:404-413)
  per LayerScaleBlock (patch tokens only, no class token yet):
      LN -> qkv GEMM -> talking-heads attention (proj_l / softmax / proj_w fused, one kernel)
         -> proj GEMM with gamma_1 folded into its weights and the shortcut in its epilogue
      LN -> fc1 GEMM (+erf GELU) -> fc2 GEMM with gamma_2 folded in, + shortcut       (cait.py:311-326)
  class token written in front of the patch tokens (one copy + one broadcast)          (cait.py:424-426)
  per LayerScaleBlockClassAttention: only token row 0 changes, so everything behind the LayerNorm runs
  on ONE row per image and updates that row of the token tensor in place:
      LN(all tokens) -> [k | v] GEMM (both kernels side by side) ; q GEMM on row 0 (scale folded in)
         -> class attention -> proj GEMM (gamma_1 folded, + x[:, 0]) -> row 0
      LN(row 0) -> fc1 -> fc2 (gamma_2 folded, + row 0) -> row 0                        (cait.py:186-202)
  final LN on row 0 only, head GEMM                                                       (cait.py:433-445)
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from ..layers.transformers import interpolate_pos_embeddings
from ..models.config import ModelConfig
from ..models.model import Model, WeightSpec
from ..models.registry import register_model
from ..utils.constants import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD

__all__ = ["CaiT", "CaiTConfig"]

_LN_EPS = {"layer_norm": 1e-5, "layer_norm_eps_1e-6": 1e-6}  # layers/factory.py:42-50


@dataclass
class CaiTConfig(ModelConfig):
    nb_classes: int = 1000
    in_channels: int = 3
    input_size: Tuple[int, int] = (224, 224)
    patch_size: int = 16
    embed_dim: int = 768
    nb_blocks: int = 12
    nb_heads: int = 12
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    # Regularization (identity at inference)
    drop_rate: float = 0.0
    drop_path_rate: float = 0.0
    attn_drop_rate: float = 0.0
    # Other parameters
    norm_layer: str = "layer_norm_eps_1e-6"
    act_layer: str = "gelu"
    init_scale: float = 1e-4
    # Parameters for inference
    interpolate_input: bool = False
    crop_pct: float = 1.0
    interpolation: str = "bicubic"
    mean: Tuple[float, float, float] = IMAGENET_DEFAULT_MEAN
    std: Tuple[float, float, float] = IMAGENET_DEFAULT_STD
    # Weight transfer
    first_conv: str = "patch_embed/proj"
    classifier: str = "head"

    @property
    def grid_size(self) -> Tuple[int, int]:
        return (self.input_size[0] // self.patch_size, self.input_size[1] // self.patch_size)

    @property
    def nb_patches(self) -> int:
        return self.grid_size[0] * self.grid_size[1]


class CaiT(Model):
    cfg_class = CaiTConfig

    def __init__(self, cfg: CaiTConfig, *args, **kwargs):
        if cfg.norm_layer not in _LN_EPS:
            raise ValueError(f"Unknown normalization layer: {cfg.norm_layer}")
        self.nb_features = cfg.embed_dim
        super().__init__(cfg, *args, **kwargs)

    # -- variables (cait.py:104-115, 158-191, 218-231, 273-306, 339-384) ------------------------------
    def weight_specs(self):
        c = self.cfg
        D, Hd, nh = c.embed_dim, int(c.embed_dim * c.mlp_ratio), c.nb_heads
        s = OrderedDict()
        s["patch_embed/proj/kernel"] = WeightSpec((c.patch_size, c.patch_size, c.in_channels, D), "conv")
        s["patch_embed/proj/bias"] = WeightSpec((D,), "bias")

        def common(p):
            s[p + "norm2/gamma"] = WeightSpec((D,), "gamma")
            s[p + "norm2/beta"] = WeightSpec((D,), "beta")
            s[p + "mlp/fc1/kernel"] = WeightSpec((D, Hd), "dense")
            s[p + "mlp/fc1/bias"] = WeightSpec((Hd,), "bias")
            s[p + "mlp/fc2/kernel"] = WeightSpec((Hd, D), "dense")
            s[p + "mlp/fc2/bias"] = WeightSpec((D,), "bias")
            s[p + "gamma_1"] = WeightSpec((D,), "scale", init=str(c.init_scale))
            s[p + "gamma_2"] = WeightSpec((D,), "scale", init=str(c.init_scale))

        for j in range(c.nb_blocks):
            p = f"blocks/{j}/"
            s[p + "norm1/gamma"] = WeightSpec((D,), "gamma")
            s[p + "norm1/beta"] = WeightSpec((D,), "beta")
            s[p + "attn/qkv/kernel"] = WeightSpec((D, 3 * D), "dense")
            if c.qkv_bias:
                s[p + "attn/qkv/bias"] = WeightSpec((3 * D,), "bias")
            s[p + "attn/proj/kernel"] = WeightSpec((D, D), "dense")
            s[p + "attn/proj/bias"] = WeightSpec((D,), "bias")
            for nm in ("proj_l", "proj_w"):
                s[p + f"attn/{nm}/kernel"] = WeightSpec((nh, nh), "dense")
                s[p + f"attn/{nm}/bias"] = WeightSpec((nh,), "bias")
            common(p)
        for j in range(2):
            p = f"blocks_token_only/{j}/"
            s[p + "norm1/gamma"] = WeightSpec((D,), "gamma")
            s[p + "norm1/beta"] = WeightSpec((D,), "beta")
            for nm in ("q", "k", "v"):
                s[p + f"attn/{nm}/kernel"] = WeightSpec((D, D), "dense")
                if c.qkv_bias:
                    s[p + f"attn/{nm}/bias"] = WeightSpec((D,), "bias")
            s[p + "attn/proj/kernel"] = WeightSpec((D, D), "dense")
            s[p + "attn/proj/bias"] = WeightSpec((D,), "bias")
            common(p)
        s["norm/gamma"] = WeightSpec((D,), "gamma")
        s["norm/beta"] = WeightSpec((D,), "beta")
        if c.nb_classes > 0:
            s["head/kernel"] = WeightSpec((D, c.nb_classes), "dense")
            s["head/bias"] = WeightSpec((c.nb_classes,), "bias")
        s["cls_token"] = WeightSpec((1, 1, D), "token")
        s["pos_embed"] = WeightSpec((1, c.nb_patches, D), "pos")
        return s

    @property
    def transform_weights(self):
        """cait.py:91-93, 386-392: position embeddings follow the target's patch grid (no token rows)."""
        return {"pos_embed": CaiT.transform_pos_embed}

    def transform_pos_embed(self, src_weights, target_cfg):
        return interpolate_pos_embeddings(src_weights, self.cfg.grid_size, target_cfg.grid_size, 0)

    @property
    def feature_names(self) -> List[str]:
        c = self.cfg
        return (["patch_embedding"] + [f"block_{j}" for j in range(c.nb_blocks)] + ["features_cls_token"]
                + [f"block_cls_token_{j}" for j in range(2)] + ["features_all", "features", "logits"])

    # -- lowering ------------------------------------------------------------------------------------
    def lower(self, b, H, W, want_features):
        c = self.cfg
        grid = (H // c.patch_size, W // c.patch_size)
        pos = b.wget("pos_embed")
        if (H, W) != tuple(c.input_size):
            if not c.interpolate_input:
                raise ValueError(
                    f"{c.name} was built for {c.input_size} inputs; got {(H, W)}. Create the model with "
                    "interpolate_input=True to resize the position embeddings (cait.py:407-415).")
            pos = interpolate_pos_embeddings(pos, c.grid_size, grid, 0)   # the class token has no position
        eps = _LN_EPS[c.norm_layer]
        D, nh, N = c.embed_dim, c.nb_heads, grid[0] * grid[1]
        scale = (D // nh) ** -0.5
        x = b.image_input(H, W, c.in_channels)
        pos_const = b.act_const(pos[0], "pos_embed")
        x = b.conv(x, "patch_embed/proj/kernel", stride=c.patch_size, padding=0, bias="patch_embed/proj/bias",
                   flatten=True, res_const=pos_const, res_mod=N,
                   cite="layers/transformers.py:164-170 + cait.py:404-413", name="tokens")
        if want_features:
            b.p.mark_output("patch_embedding", x)

        for j in range(c.nb_blocks):
            p = f"blocks/{j}/"
            qkv = b.ln_dense(x, p + "norm1", eps, p + "attn/qkv/kernel", p + "attn/qkv/bias" if c.qkv_bias else None,
                             cite_ln="cait.py:313", cite="cait.py:236")
            a = b.talking_heads_attention(qkv, nh, scale, p + "attn", cite="cait.py:237-256")
            x = b.dense(a, p + "attn/proj/kernel", p + "attn/proj/bias", out_scale=p + "gamma_1", residual=x,
                        cite="cait.py:258,314-317")
            hdn = b.ln_dense(x, p + "norm2", eps, p + "mlp/fc1/kernel", p + "mlp/fc1/bias", act=c.act_layer,
                             cite_ln="cait.py:320", cite="transformers.py:209-210")
            x = b.dense(hdn, p + "mlp/fc2/kernel", p + "mlp/fc2/bias", out_scale=p + "gamma_2", residual=x,
                        cite="transformers.py:212, cait.py:322-325")
            if want_features:
                b.p.mark_output(f"block_{j}", x)

        # tf.concat((cls_token, x), axis=1): the class token never saw pos_embed (cait.py:424-426)
        xc = b.empty(N + 1, D, name="tokens+cls")
        b.token_rows(xc, b.wget("cls_token")[0], cite="cait.py:424-425")
        b.copy_rows(x, xc, 1, cite="cait.py:426")

        def snapshot(name):
            # the class-token blocks update xc in place; a returned feature needs its own copy
            f = b.empty(N + 1, D, name=name)
            b.copy_rows(xc, f, 0)
            b.p.mark_output(name, f)

        if want_features:
            snapshot("features_cls_token")
        for j in range(2):
            p = f"blocks_token_only/{j}/"
            u = b.layernorm(xc, p + "norm1", eps, cite="cait.py:188")
            # k and v layers of ClassAttention as one GEMM; q only for the class-token row, pre-scaled
            kvk = b.define(p + "attn/kv/kernel", np.concatenate([b.wget(p + "attn/k/kernel"), b.wget(p + "attn/v/kernel")], 1))
            kvb = qb = None
            qk = b.define(p + "attn/q_scaled/kernel", b.wget(p + "attn/q/kernel") * scale)
            if c.qkv_bias:
                kvb = b.define(p + "attn/kv/bias", np.concatenate([b.wget(p + "attn/k/bias"), b.wget(p + "attn/v/bias")]))
                qb = b.define(p + "attn/q_scaled/bias", b.wget(p + "attn/q/bias") * scale)
            kv = b.dense(u, kvk, kvb, cite="cait.py:129-135")
            q = b.dense(u, qk, qb, row_select=(0, 1), cite="cait.py:122-127")
            ca = b.class_attention(q, kv, nh, cite="cait.py:137-143")
            b.dense(ca, p + "attn/proj/kernel", p + "attn/proj/bias", out_scale=p + "gamma_1", residual=xc,
                    residual_row=0, out=xc, out_row=0, cite="cait.py:145,189-191")
            y = b.layernorm(xc, p + "norm2", eps, row_select=(0, 1), cite="cait.py:194")
            hdn = b.dense(y, p + "mlp/fc1/kernel", p + "mlp/fc1/bias", act=c.act_layer, cite="transformers.py:209-210")
            b.dense(hdn, p + "mlp/fc2/kernel", p + "mlp/fc2/bias", out_scale=p + "gamma_2", residual=xc,
                    residual_row=0, out=xc, out_row=0, cite="cait.py:195-198")
            if want_features:
                snapshot(f"block_cls_token_{j}")
        if want_features:
            allf = b.layernorm(xc, "norm", eps, cite="cait.py:433")
            b.p.mark_output("features_all", allf)
        features = b.layernorm(xc, "norm", eps, row_select=(0, 1), cite="cait.py:433,435", name="norm[cls]")
        b.p.mark_output("features", features)
        if c.nb_classes > 0:
            logits = b.dense(features, "head/kernel", "head/bias", out_f32=True, cite="cait.py:444")
            b.p.mark_output("logits", logits)
        else:
            b.p.mark_output("logits", features)

    def _shape_output(self, name, v):
        if name in ("features", "logits"):
            return v.reshape(v.shape[0], -1)
        return v


# ---------------------------------------------------------------------------------------
# registrations (reference cait.py:448-607)
# ---------------------------------------------------------------------------------------
def _register(name, res, dim, blocks, heads, init_scale):
    cfg = CaiTConfig(name=name, url="[timm]", input_size=(res, res), patch_size=16, embed_dim=dim, nb_blocks=blocks,
                     nb_heads=heads, init_scale=init_scale)

    def fn():
        return CaiT, cfg
    fn.__name__ = name
    fn.__qualname__ = name
    fn.__module__ = __name__
    fn.__doc__ = f"{name} (reference tfimm/architectures/cait.py)"
    globals()[name] = register_model(fn)


for _row in [
    ("cait_xxs24_224", 224, 192, 24, 4, 1e-5),
    ("cait_xxs24_384", 384, 192, 24, 4, 1e-5),
    ("cait_xxs36_224", 224, 192, 36, 4, 1e-5),
    ("cait_xxs36_384", 384, 192, 36, 4, 1e-5),
    ("cait_xs24_384", 384, 288, 24, 6, 1e-5),
    ("cait_s24_224", 224, 384, 24, 8, 1e-5),
    ("cait_s24_384", 384, 384, 24, 8, 1e-5),
    ("cait_s36_384", 384, 384, 36, 8, 1e-6),
    ("cait_m36_384", 384, 768, 36, 16, 1e-6),
    ("cait_m48_448", 448, 768, 48, 16, 1e-6),
]:
    _register(*_row)
del _row
