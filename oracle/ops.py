"""CPU restatement of the TensorFlow/Keras 2.12 ops the tfimm forward path calls.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import anything under oracle/.  The product path
(tensorflow-image-models_amd/) never does and fails loudly without its HIP library.

Parity status: TensorFlow is a third-party, un-vendored dependency of the reference
(tensorflow 2.12.0, poetry.lock:1405-1406) and cannot be installed here, so these functions
restate the published semantics of each op on top of torch-CPU fp32
tensors, NHWC / HWIO layouts exactly as the reference uses them.  They are pinned by
tests/test_oracle_ops.py against hand-computed known answers and independent numpy
formulations; the model-level restatements in this package are additionally pinned by
running the reference's OWN model code (/root/reference/tfimm, unmodified) over these ops
through the stand-in TensorFlow of oracle/tf_shim: oracle/tools/make_reference_golden.py writes
its outputs to tests/golden/forward_golden.npz and tests/test_golden.py holds every restatement
to them at 1e-5 (see oracle/README.md for what that does and does not pin).

Every function cites the reference call sites it serves.
"""
import math
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

T = torch.Tensor


def as_t(x) -> T:
    if isinstance(x, torch.Tensor):
        return x.float()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32))


# ---- padding ------------------------------------------------------------------------------
def zero_pad2d(x: T, pad: Union[int, Tuple[int, int], Tuple[Tuple[int, int], Tuple[int, int]]]) -> T:
    """tf.keras.layers.ZeroPadding2D on NHWC (resnet.py:229,505; layers/conv.py:80-88)."""
    if isinstance(pad, int):
        pt = pb = pl = pr = pad
    elif isinstance(pad[0], int):
        pt = pb = pad[0]
        pl = pr = pad[1]
    else:
        (pt, pb), (pl, pr) = pad
    return F.pad(x, (0, 0, pl, pr, pt, pb))


def same_pad_amounts(size: int, k: int, s: int, d: int = 1) -> Tuple[int, int]:
    """TF padding="same": out = ceil(in/s); total = max((out-1)*s + k_eff - in, 0);
    before = total // 2, after = total - before."""
    k_eff = (k - 1) * d + 1
    out = -(-size // s)
    total = max((out - 1) * s + k_eff - size, 0)
    return total // 2, total - total // 2


# ---- convolutions -----------------------------------------------------------------------------
def _pair(v) -> Tuple[int, int]:
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def conv2d(x: T, kernel: T, bias: Optional[T] = None, stride=1, padding: str = "valid",
           groups: int = 1, dilation=1) -> T:
    """tf.keras.layers.Conv2D: NHWC input, HWIO kernel, cross-correlation
    (vit/resnet/efficientnet conv call sites; transformers.py:155-163)."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    sh, sw = _pair(stride)
    dh, dw = _pair(dilation)
    if padding == "same":
        pt, pb = same_pad_amounts(x.shape[1], kh, sh, dh)
        pl, pr = same_pad_amounts(x.shape[2], kw, sw, dw)
        x = zero_pad2d(x, ((pt, pb), (pl, pr)))
    elif padding != "valid":
        raise ValueError(padding)
    w = kernel.permute(3, 2, 0, 1).contiguous()  # HWIO -> OIHW
    y = F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride=(sh, sw), dilation=(dh, dw), groups=groups)
    return y.permute(0, 2, 3, 1).contiguous()


def depthwise_conv2d(x: T, kernel: T, bias: Optional[T] = None, stride=1,
                     padding: str = "valid", dilation=1) -> T:
    """tf.keras.layers.DepthwiseConv2D: kernel (kh, kw, C, 1) (layers/conv.py:91-148)."""
    kh, kw, c, mult = kernel.shape
    assert mult == 1
    sh, sw = _pair(stride)
    dh, dw = _pair(dilation)
    if padding == "same":
        pt, pb = same_pad_amounts(x.shape[1], kh, sh, dh)
        pl, pr = same_pad_amounts(x.shape[2], kw, sw, dw)
        x = zero_pad2d(x, ((pt, pb), (pl, pr)))
    w = kernel.permute(2, 3, 0, 1).contiguous()  # (C, 1, kh, kw)
    y = F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride=(sh, sw), dilation=(dh, dw), groups=c)
    return y.permute(0, 2, 3, 1).contiguous()


def dense(x: T, kernel: T, bias: Optional[T] = None) -> T:
    """tf.keras.layers.Dense: x @ kernel (in, out) + bias on the last axis."""
    y = x @ kernel
    return y if bias is None else y + bias


# ---- normalisation ------------------------------------------------------------------------------
def layer_norm(x: T, gamma: T, beta: T, eps: float) -> T:
    """Keras LayerNormalization, non-fused path (eps < 1.001e-5): two-pass population
    moments over the last axis, y = x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


def batch_norm(x: T, gamma: T, beta: T, mean: T, var: T, eps: float) -> T:
    """Keras BatchNormalization at training=False: gamma*(x-mean)*rsqrt(var+eps)+beta."""
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def group_norm(x: T, gamma: T, beta: T, groups: int, eps: float) -> T:
    """layers/norm.py:37-107 group_normalize on N..C input: reshape to N..GS, population moments over every axis
    but N and G (tf.nn.moments), tf.nn.batch_normalization with per-channel gamma / beta."""
    shape = x.shape
    c = shape[-1]
    xg = x.reshape(*shape[:-1], groups, c // groups)
    dims = tuple(range(1, xg.dim() - 2)) + (xg.dim() - 1,)
    mean = xg.mean(dim=dims, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=dims, keepdim=True)
    inv = torch.rsqrt(var + eps) * gamma.reshape(groups, c // groups)
    y = xg * inv + (beta.reshape(groups, c // groups) - mean * inv)
    return y.reshape(shape)


# ---- activations (layers/factory.py:6-13) ---------------------------------------------------------
def activation(x: T, name: str) -> T:
    if name in ("linear", "", None):
        return x
    if name == "relu":
        return torch.relu(x)
    if name == "relu6":
        return torch.clamp(x, 0.0, 6.0)
    if name == "gelu":  # keras gelu(approximate=False): exact erf
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if name == "swish":
        return x * torch.sigmoid(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "tanh":
        return torch.tanh(x)
    raise ValueError(f"Unknown activation: {name}.")


def softmax(x: T, axis: int = -1) -> T:
    return torch.softmax(x, dim=axis)


# ---- pooling ----------------------------------------------------------------------------------------
def max_pool2d(x: T, k: int, stride: int, padding: str = "valid") -> T:
    """tf.keras.layers.MaxPool2D(pool_size=k, strides=stride), NHWC (resnet.py:539).  "same" pads with
    -inf, i.e. border windows take the maximum over their valid elements."""
    if padding == "same":
        pt, pb = same_pad_amounts(x.shape[1], k, stride)
        pl, pr = same_pad_amounts(x.shape[2], k, stride)
        x = F.pad(x, (0, 0, pl, pr, pt, pb), value=float("-inf"))
    y = F.max_pool2d(x.permute(0, 3, 1, 2), k, stride)
    return y.permute(0, 2, 3, 1).contiguous()


def avg_pool2d_valid(x: T, k: int, stride: int) -> T:
    """AveragePooling2D(padding="valid")."""
    return F.avg_pool2d(x.permute(0, 3, 1, 2), k, stride).permute(0, 2, 3, 1).contiguous()


def avg_pool2d_same(x: T, k: int, stride: int) -> T:
    """AveragePooling2D(padding="same"): windows are clipped at the border and the average
    is taken over the VALID elements only (resnet.py:299-301, downsample_avg)."""
    pt, pb = same_pad_amounts(x.shape[1], k, stride)
    pl, pr = same_pad_amounts(x.shape[2], k, stride)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    ones = F.pad(torch.ones_like(x[..., :1]).permute(0, 3, 1, 2), (pl, pr, pt, pb))
    s = F.avg_pool2d(xp, k, stride) * (k * k)
    n = F.avg_pool2d(ones, k, stride) * (k * k)
    return (s / n).permute(0, 2, 3, 1).contiguous()


def blur_pool2d(x: T, stride: int, kernel_size: int = 3) -> T:
    """layers/blurpool.py:5-66 BlurPool2D: tf.pad(REFLECT) by (k + stride) // 2 - 1, then the binomial depthwise
    filter ([1 2 1] x [1 2 1] / 16 for k = 3) at `stride`, VALID."""
    assert kernel_size == 3
    p = (kernel_size + stride) // 2 - 1
    xp = F.pad(x.permute(0, 3, 1, 2), (p, p, p, p), mode="reflect") if p else x.permute(0, 3, 1, 2)
    c = x.shape[-1]
    bk = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0
    y = F.conv2d(xp, bk.reshape(1, 1, 3, 3).repeat(c, 1, 1, 1), stride=stride, groups=c)
    return y.permute(0, 2, 3, 1).contiguous()


def global_avg_pool(x: T) -> T:
    """GlobalAveragePooling2D (NHWC -> NC) / GlobalAveragePooling1D (NLC -> NC)."""
    return x.mean(dim=tuple(range(1, x.dim() - 1)))


# ---- tensor shuffles --------------------------------------------------------------------------------
def roll(x: T, shift: Sequence[int], axis: Sequence[int]) -> T:
    """tf.roll: out[i] = in[(i - shift) mod n]  (swin.py:299,313)."""
    return torch.roll(x, shifts=tuple(shift), dims=tuple(axis))


# ---- image resize -----------------------------------------------------------------------------
def resize_bicubic_tf(x: T, size: Tuple[int, int]) -> T:
    """tf.image.resize(method="bicubic", antialias=False) on NHWC (layers/transformers.py:38-42),
    restated tap by tap from TensorFlow's resize_bicubic_op (half-pixel centres; Keys kernel A = -0.5
    read from a 1024-entry table at the ROUNDED fractional offset; taps outside the image weigh 0 and
    the rest are renormalised).  Deliberately written as plain loops, independently of the engine's
    vectorised host version (tfimm/layers/transformers.py)."""
    A = -0.5
    TAB = 1024

    def near(i):
        t = np.float32(i) / np.float32(TAB)
        return np.float32(((A + 2) * t - (A + 3)) * t * t + 1)

    def far(i):
        t = np.float32(i) / np.float32(TAB) + np.float32(1)
        return np.float32(((A * t - 5 * A) * t + 8 * A) * t - 4 * A)

    def taps(n_in, n_out):
        out = []
        scale = np.float32(n_in) / np.float32(n_out)
        for o in range(n_out):
            loc = (np.float32(o) + np.float32(0.5)) * scale - np.float32(0.5)
            base = math.floor(float(loc))
            off = int(np.rint((loc - np.float32(base)) * TAB))
            cand = [(base - 1, far(off)), (base, near(off)), (base + 1, near(TAB - off)), (base + 2, far(TAB - off))]
            kept = [(i, w) for i, w in cand if 0 <= i < n_in]
            tot = sum(w for _, w in kept)
            out.append([(i, np.float32(w / tot)) for i, w in kept])
        return out

    x = as_t(x)
    B, H, W, C = x.shape
    ty, tx = taps(H, size[0]), taps(W, size[1])
    rows = torch.zeros(B, size[0], W, C)
    for oy, tl in enumerate(ty):
        for i, w in tl:
            rows[:, oy] += float(w) * x[:, i]
    out = torch.zeros(B, size[0], size[1], C)
    for ox, tl in enumerate(tx):
        for i, w in tl:
            out[:, :, ox] += float(w) * rows[:, :, i]
    return out


def interpolate_pos_embeddings(pos_embed: T, src_grid, tgt_grid, nb_tokens: int = 0) -> T:
    """layers/transformers.py:13-47."""
    pos_embed = as_t(pos_embed)
    if tuple(src_grid) == tuple(tgt_grid):
        return pos_embed
    g = pos_embed[:, nb_tokens:].reshape(1, src_grid[0], src_grid[1], -1)
    g = resize_bicubic_tf(g, tgt_grid).reshape(1, tgt_grid[0] * tgt_grid[1], -1)
    return torch.cat((pos_embed[:, :nb_tokens], g), dim=1)
