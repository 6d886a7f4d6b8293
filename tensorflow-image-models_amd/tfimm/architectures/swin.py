"""Swin Transformer on the MI355X engine.

Behavioural mirror of reference tfimm/architectures/swin.py (SwinTransformerConfig :28-69,
window_partition/_reverse :72-108, WindowAttention :111-198, SwinTransformerBlock :201-327,
PatchMerging :330-362, SwinTransformerStage :365-407, SwinTransformer :410-517,
registrations :520-679).  Lowering of one block (swin.py:287-327):

  LN -> QKV GEMM on tokens in their NATURAL (y, x) order (the projection is per token, so
  tf.roll / window_partition need not run before it) -> fused window attention that applies
  roll(-s), window_partition, + relative-position bias, + shift mask, softmax, .V,
  window_reverse and roll(+s) purely as load/store index maps -> proj GEMM (+residual)
  -> LN -> fc1 GEMM (+erf GELU) -> fc2 GEMM (+residual)

PatchMerging (swin.py:348-362) = one gather+LayerNorm launch + one bias-free GEMM.
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from ..models.config import ModelConfig
from ..models.model import Model, WeightSpec
from ..models.registry import register_model
from ..utils.constants import IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD

__all__ = ["SwinTransformer", "SwinTransformerConfig"]

_LN_EPS = {"layer_norm": 1e-5, "layer_norm_eps_1e-6": 1e-6}


@dataclass
class SwinTransformerConfig(ModelConfig):
    nb_classes: int = 1000
    in_channels: int = 3
    input_size: Tuple[int, int] = (224, 224)
    patch_size: int = 4
    embed_dim: int = 96
    nb_blocks: Tuple = (2, 2, 6, 2)
    nb_heads: Tuple = (3, 6, 12, 24)
    window_size: int = 7
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    # Regularization
    drop_rate: float = 0.0
    attn_drop_rate: float = 0.0
    drop_path_rate: float = 0.1
    # Other parameters
    norm_layer: str = "layer_norm"
    act_layer: str = "gelu"
    patch_norm: bool = True
    # Parameters for inference
    interpolate_input: bool = False
    crop_pct: float = 0.9
    interpolation: str = "bicubic"
    # Preprocessing
    mean: Tuple[float, float, float] = IMAGENET_DEFAULT_MEAN
    std: Tuple[float, float, float] = IMAGENET_DEFAULT_STD
    # Weight transfer
    first_conv: str = "patch_embed/proj"
    classifier: str = "head"

    @property
    def patch_resolution(self):
        return (self.input_size[0] // self.patch_size, self.input_size[1] // self.patch_size)

    @property
    def nb_patches(self):
        return self.patch_resolution[0] * self.patch_resolution[1]


def relative_position_index(ws: int) -> np.ndarray:
    """(ws*ws, ws*ws) int index into the (2ws-1)^2 bias table (swin.py:143-152)."""
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).astype(np.int64)


class SwinTransformer(Model):
    cfg_class = SwinTransformerConfig
    keys_to_ignore_on_load = ("attn_mask", "relative_position_index")

    def _stage_dims(self):
        c = self.cfg
        for i in range(len(c.nb_blocks)):
            res = (c.patch_resolution[0] // 2 ** i, c.patch_resolution[1] // 2 ** i)
            yield i, res, int(c.embed_dim * 2 ** i), c.nb_heads[i], c.nb_blocks[i]

    def weight_specs(self):
        c = self.cfg
        s = OrderedDict()
        s["patch_embed/proj/kernel"] = WeightSpec((c.patch_size, c.patch_size, c.in_channels, c.embed_dim), "conv")
        s["patch_embed/proj/bias"] = WeightSpec((c.embed_dim,), "bias")
        s["patch_embed/norm/gamma"] = WeightSpec((c.embed_dim,), "gamma")
        s["patch_embed/norm/beta"] = WeightSpec((c.embed_dim,), "beta")
        nst = len(c.nb_blocks)
        ws = c.window_size
        for i, res, D, nh, nb in self._stage_dims():
            Hd = int(D * c.mlp_ratio)
            for j in range(nb):
                p = f"layers/{i}/blocks/{j}/"
                s[p + "norm1/gamma"] = WeightSpec((D,), "gamma")
                s[p + "norm1/beta"] = WeightSpec((D,), "beta")
                s[p + "attn/relative_position_bias_table"] = WeightSpec(((2 * ws - 1) ** 2, nh), "table")
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
            if i < nst - 1:
                s[f"layers/{i}/downsample/norm/gamma"] = WeightSpec((4 * D,), "gamma")
                s[f"layers/{i}/downsample/norm/beta"] = WeightSpec((4 * D,), "beta")
                s[f"layers/{i}/downsample/reduction/kernel"] = WeightSpec((4 * D, 2 * D), "dense")
        Dl = int(c.embed_dim * 2 ** (nst - 1))
        s["norm/gamma"] = WeightSpec((Dl,), "gamma")
        s["norm/beta"] = WeightSpec((Dl,), "beta")
        if c.nb_classes > 0:
            s["head/kernel"] = WeightSpec((Dl, c.nb_classes), "dense")
            s["head/bias"] = WeightSpec((c.nb_classes,), "bias")
        self.nb_features = Dl
        return s

    @property
    def feature_names(self) -> List[str]:
        names = ["patch_embedding"]
        k = 0
        for j, nb in enumerate(self.cfg.nb_blocks):
            for _ in range(nb):
                names.append(f"block_{k}")
                k += 1
            names.append(f"stage_{j}")
        return names + ["features_all", "features", "logits"]

    def lower(self, b, H, W, want_features):
        c = self.cfg
        if (H, W) != tuple(c.input_size):
            raise NotImplementedError("Swin only runs at its configured input size (tests/models/test_factory.py:24-27).")
        eps = _LN_EPS[c.norm_layer]
        x = b.image_input(H, W, c.in_channels)
        x = b.conv(x, "patch_embed/proj/kernel", stride=c.patch_size, padding=0, bias="patch_embed/proj/bias",
                   cite="layers/transformers.py:164-165")
        x = b.layernorm(x, "patch_embed/norm", eps, cite="layers/transformers.py:172")
        if want_features:
            b.p.mark_output("patch_embedding", x)
        nst = len(c.nb_blocks)
        k = 0
        for i, res, D, nh, nb in self._stage_dims():
            assert (x.H, x.W) == res and x.C == D
            for j in range(nb):
                p = f"layers/{i}/blocks/{j}/"
                ws = c.window_size
                shift = 0 if j % 2 == 0 else c.window_size // 2       # swin.py:387
                if min(res) <= c.window_size:                         # swin.py:221-223
                    shift, ws = 0, min(res)
                if ws != c.window_size:
                    raise NotImplementedError(
                        "stage resolution smaller than window_size: the reference's WindowAttention reshapes with "
                        "cfg.window_size (swin.py:179-182) and cannot run this case either.")
                table = b.wget(p + "attn/relative_position_bias_table")
                n = ws * ws
                bias = table[relative_position_index(ws).reshape(-1)].reshape(n, n, nh).transpose(2, 0, 1)
                qkv = b.ln_dense(x, p + "norm1", eps, p + "attn/qkv/kernel", p + "attn/qkv/bias" if c.qkv_bias else None,
                                 cite_ln="swin.py:295", cite="swin.py:167")
                a = b.attention(qkv, nh, (D // nh) ** -0.5, window=ws, shift=shift, res=res,
                                rel_bias=np.ascontiguousarray(bias), cite="swin.py:299-313 + 168-195", name=p + "attn")
                x = b.dense(a, p + "attn/proj/kernel", p + "attn/proj/bias", residual=x, cite="swin.py:196,318")
                y = b.mlp_fused(x, p + "norm2", eps, p + "mlp/fc1", p + "mlp/fc2", act=c.act_layer,
                                cite="swin.py:322-325, transformers.py:208-214")
                if y is None:
                    hdn = b.ln_dense(x, p + "norm2", eps, p + "mlp/fc1/kernel", p + "mlp/fc1/bias", act=c.act_layer,
                                     cite_ln="swin.py:322", cite="transformers.py:209-210")
                    y = b.dense(hdn, p + "mlp/fc2/kernel", p + "mlp/fc2/bias", residual=x,
                                cite="transformers.py:212, swin.py:325")
                x = y
                x.H, x.W = res
                if want_features:
                    b.p.mark_output(f"block_{k}", x)
                k += 1
            if i < nst - 1:
                m = b.patch_merge_ln(x, f"layers/{i}/downsample/norm", eps, cite="swin.py:352-360")
                x = b.dense(m, f"layers/{i}/downsample/reduction/kernel", None, cite="swin.py:361")
                x.H, x.W = res[0] // 2, res[1] // 2
            if want_features:
                b.p.mark_output(f"stage_{i}", x)
        x = b.layernorm(x, "norm", eps, cite="swin.py:504")
        if want_features:
            b.p.mark_output("features_all", x)
        pooled = b.mean_rows(x, cite="swin.py:506")
        b.p.mark_output("features", pooled)
        if c.nb_classes > 0:
            logits = b.dense(pooled, "head/kernel", "head/bias", out_f32=True, cite="swin.py:515")
        else:
            logits = pooled
        b.p.mark_output("logits", logits)

    def _shape_output(self, name, v):
        if v.dim() == 4:                       # token tensors are (B, L, C) in the reference
            return v.reshape(v.shape[0], -1, v.shape[-1])
        return v


def _swin(name, embed, blocks, heads, res=224, window=7, **kw):
    cfg = dict(name=name, url="[timm]", embed_dim=embed, nb_blocks=blocks, nb_heads=heads, window_size=window)
    if res != 224:
        cfg.update(input_size=(res, res), crop_pct=1.0)
    cfg.update(kw)
    return SwinTransformerConfig(**cfg)


def _register(cfg):
    def fn():
        return SwinTransformer, cfg
    fn.__name__ = fn.__qualname__ = cfg.name
    fn.__module__ = __name__
    fn.__doc__ = f"{cfg.name} (reference tfimm/architectures/swin.py)"
    globals()[cfg.name] = register_model(fn)


_B, _L = ((2, 2, 18, 2), (4, 8, 16, 32)), ((2, 2, 18, 2), (6, 12, 24, 48))
for _cfg in [
    _swin("swin_tiny_patch4_window7_224", 96, (2, 2, 6, 2), (3, 6, 12, 24)),
    _swin("swin_small_patch4_window7_224", 96, (2, 2, 18, 2), (3, 6, 12, 24)),
    _swin("swin_base_patch4_window7_224", 128, *_B),
    _swin("swin_base_patch4_window12_384", 128, *_B, res=384, window=12),
    _swin("swin_large_patch4_window7_224", 192, *_L),
    _swin("swin_large_patch4_window12_384", 192, *_L, res=384, window=12),
    _swin("swin_base_patch4_window7_224_in22k", 128, *_B, nb_classes=21841),
    _swin("swin_base_patch4_window12_384_in22k", 128, *_B, res=384, window=12, nb_classes=21841),
    _swin("swin_large_patch4_window7_224_in22k", 192, *_L, nb_classes=21841),
    _swin("swin_large_patch4_window12_384_in22k", 192, *_L, res=384, window=12, nb_classes=21841),
]:
    _register(_cfg)
del _cfg
