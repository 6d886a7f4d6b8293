"""Op-level parity cases: every HIP kernel vs. the CPU oracle ops on identical bf16-valued inputs.

Each case returns ``(err, tol)`` with err = max|hip - ref| / (max|ref| + 1e-6).  Inputs are
rounded to bf16 first and the reference is evaluated in fp32 on those values, so the only
differences are accumulation order and the final bf16 rounding of the output
(bf16 has 8 significant bits: 2^-8 = 3.9e-3; tolerance 1e-2 rel-to-max for bf16 outputs,
1e-3 for fp32 outputs).  Used by tests/test_gpu_ops.py (pytest -m gpu) and by
tools/gpu_selftest.py (prints the full table without stopping at the first failure).
"""
import math

import numpy as np
import torch

from oracle import ops as O
from tfimm.engine import pack

TOL_BF16 = 1e-2
TOL_F32 = 1e-3


def _rng(seed):
    return np.random.default_rng(seed)


def _bf(a):
    """round fp32 numpy -> bf16-representable fp32 numpy"""
    return pack.bf16_bits_to_f32(pack.to_bf16_bits(np.asarray(a, dtype=np.float32))).reshape(np.shape(a))


_TIGHT = False     # set by the tight_* cases at the bottom of this file: _err then measures element-wise, in bf16 ulps


def _err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if not np.all(np.isfinite(got)):
        return float("inf")
    if _TIGHT:
        return _err_ulp(got, ref)
    return float(np.max(np.abs(got - ref)) / (np.max(np.abs(ref)) + 1e-6))


def _err_ulp(got, ref):
    """Element-wise bar of the tight cases: every stored bf16 value within 2 bf16 ulps OF ITS OWN REFERENCE VALUE (ulp =
    2^(floor(log2 |ref|) - 7)), plus a floor of 2^-14 of the tensor's rms for values so small that fp32 accumulation noise
    and the 1.6e-5 of the GELU polynomial exceed their ulp.  Returned: max |got - ref| / allowed (<= 1 passes; a launch with a
    single rounding measures 0.25 = half an ulp here).  A wrong
    epsilon, a tanh-form GELU (up to 5e-4 off where the exact value is ~0.02) or an activation on the wrong side of the
    residual add are all several units on this scale; rel-to-max 1e-2 sees none of them."""
    a = np.abs(ref)
    ulp = np.exp2(np.floor(np.log2(np.maximum(a, 1e-30))) - 7)
    rms = np.sqrt(np.mean(ref * ref))
    err = np.abs(got - ref)
    strict = float(np.max(err / (2.0 * ulp + rms * 2.0 ** -14)))
    if _TIGHT == 1 or strict <= 1.0:
        return strict
    # _TIGHT == 2, launches with a bf16 intermediate (conv_chain, mlp_fused, expand_dwconv): an intermediate value within fp32
    # accumulation noise of a rounding boundary may round the other way than in the reference (~5e-4 of them), which moves the
    # outputs it feeds by |w| ulp(intermediate) -- many ulps of an output that happens to be near zero.  Those are isolated
    # elements: at most 0.5 % of the tensor may exceed the strict bar, and none may exceed 2 ulps + 2^-8 rms.
    over = float(np.mean(err > 2.0 * ulp + rms * 2.0 ** -14))
    loose = float(np.max(err / (2.0 * ulp + rms * 2.0 ** -8)))
    return max(over / 0.005, loose)


def _cpu(t):
    return t.float().cpu().numpy()


CASES = {}


def case(name):
    def deco(fn):
        CASES[name] = fn
        return fn
    return deco


# ---------------------------------------------------------------------------------------------
# GEMM (dense)
# ---------------------------------------------------------------------------------------------
def _gemm_case(M, K, N, *, act="", bias=True, residual=False, act_after_res=False, out_f32=False, tile=0,
               seed=0, res_mod=0, remap=None):
    import hip_ops as H
    r = _rng(seed)
    a = _bf(r.standard_normal((M, K)))
    w = _bf(r.standard_normal((K, N)) / math.sqrt(K))
    b = r.standard_normal(N).astype(np.float32) if bias else None
    nres = res_mod if res_mod else M
    res = _bf(r.standard_normal((nres, N))) if residual else None
    wt, _ = pack.pack_dense(w, None)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    if bias:
        ref = ref + b
    ref_t = torch.from_numpy(ref.astype(np.float32))
    if not act_after_res:
        ref_t = O.activation(ref_t, act)
    if residual:
        idx = np.arange(M) % nres
        ref_t = ref_t + torch.from_numpy(res[idx])
    if act_after_res:
        ref_t = O.activation(ref_t, act)
    ref = ref_t.numpy()
    out_rows = None
    if remap:
        rin, rout, roff = remap
        out_rows = (M // rin) * rout
    got = H.gemm(H.dev_bf16(a), H.dev_bits(wt), N, K, bias=None if b is None else H.dev_f32(b),
                 residual=None if res is None else H.dev_bf16(res), act=act, act_after_res=act_after_res,
                 out_f32=out_f32, tile_hint=tile, res_mod=res_mod, remap=remap, out_rows=out_rows)
    H.sync()
    got = _cpu(got)
    if remap:
        rin, rout, roff = remap
        rows = (np.arange(M) // rin) * rout + np.arange(M) % rin + roff
        untouched = np.setdiff1d(np.arange(out_rows), rows)
        assert np.all(got[untouched] == 0), "remap wrote outside its rows"
        got = got[rows]
    return _err(got, ref), (TOL_F32 if out_f32 else TOL_BF16)


# tile hints: 0 auto, 1..6 register-staged tiles, 11..16 LDS-DMA tiles, 21..27 and 29 persistent LDS-DMA tiles,
# 28 the 256x256 deep-ring schedule, 30 the 256x128 tile with two co-resident four-wave workgroups per CU (gemm_duo_kernel.h)
for _t in list(range(0, 7)) + list(range(11, 17)) + list(range(21, 31)):
    CASES[f"gemm_tile{_t:02d}_256x192x320"] = (lambda t=_t: _gemm_case(256, 192, 320, tile=t, seed=1))
    CASES[f"gemm_tile{_t:02d}_ragged_333x200x150_gelu_res"] = (
        lambda t=_t: _gemm_case(333, 200, 150, act="gelu", residual=True, tile=t, seed=2))
    CASES[f"gemm_tile{_t:02d}_600x320x520_relu_after_res_f32"] = (
        lambda t=_t: _gemm_case(600, 320, 520, act="relu", residual=True, act_after_res=True, out_f32=True, tile=t,
                                seed=3))
for _t in range(21, 31):
    # more tiles than resident workgroups: every persistent workgroup walks several tiles (ragged M, N, K)
    CASES[f"gemm_stream_multiround_tile{_t:02d}"] = (
        lambda t=_t: _gemm_case(40000, 200, 520, act="gelu", residual=True, tile=t, seed=50 + t))
    CASES[f"gemm_stream_k64_tile{_t:02d}"] = (lambda t=_t: _gemm_case(70000, 64, 256, act="relu", tile=t, seed=60 + t))
CASES["gemm_vit_qkv_394x768x2304"] = lambda: _gemm_case(394, 768, 2304, seed=3)
CASES["gemm_vit_fc2_394x3072x768_res"] = lambda: _gemm_case(394, 3072, 768, residual=True, seed=4)
CASES["gemm_head_f32_8x768x1000"] = lambda: _gemm_case(8, 768, 1000, out_f32=True, seed=5)
CASES["gemm_relu_after_res_512x64x256"] = lambda: _gemm_case(512, 64, 256, act="relu", residual=True,
                                                             act_after_res=True, seed=6)
CASES["gemm_swish_1000x24x144"] = lambda: _gemm_case(1000, 24, 144, act="swish", seed=7)
CASES["gemm_scalar_K4_17x4x12"] = lambda: _gemm_case(34, 4, 12, seed=8)
CASES["gemm_scalar_odd_70x5x7_f32"] = lambda: _gemm_case(70, 5, 7, out_f32=True, seed=9)
CASES["gemm_nobias_130x72x40"] = lambda: _gemm_case(130, 72, 40, bias=False, seed=10)
CASES["gemm_resmod_remap_392x64x96"] = lambda: _gemm_case(392, 64, 96, residual=True, res_mod=196,
                                                          remap=(196, 197, 1), seed=11)
CASES["gemm_tanh_64x128x64"] = lambda: _gemm_case(64, 128, 64, act="tanh", seed=12)
CASES["gemm_sigmoid_relu6"] = lambda: max(_gemm_case(64, 64, 64, act="sigmoid", seed=13),
                                          _gemm_case(64, 64, 64, act="relu6", seed=14))


def _se_scale_case(B, R, K, N, seed, tile=0, residual=False, bias=False):
    """SE gate folded into the projection conv: (a * gate[image]) @ W (efficientnet_blocks.py:241-248,447-449)."""
    import hip_ops as H
    r = _rng(seed)
    a = _bf(r.standard_normal((B * R, K)))
    w = _bf(r.standard_normal((K, N)) / math.sqrt(K))
    g = r.uniform(0.1, 1.0, (B, K)).astype(np.float32)
    bvec = r.standard_normal(N).astype(np.float32) if bias else None
    res = _bf(r.standard_normal((B * R, N))) if residual else None
    wt, bp = pack.pack_dense(w, bvec)
    scaled = _bf(a.reshape(B, R, K) * g[:, None, :]).reshape(B * R, K)   # kernel re-rounds A*gate to bf16
    ref = scaled.astype(np.float64) @ w.astype(np.float64)
    if bias:
        ref = ref + bvec
    if residual:
        ref = ref + res
    got = H.gemm(H.dev_bf16(a), H.dev_bits(wt), N, K, a_scale=H.dev_f32(g), rows_per_image=R, tile_hint=tile,
                 bias=None if bp is None else H.dev_f32(bp), residual=None if res is None else H.dev_bf16(res))
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


def _dual_case(B, OH, OW, K1, K2, N, stride, seed, tile=0, act="relu"):
    """A second A operand (tfimm_gemm_desc::a2, ABI v4): conv3 of a bottleneck + its 1x1 / stride-s shortcut convolution as
    ONE GEMM -- out = act(h . W3 + x[:, ::s, ::s] . Wds + b3 + bds)  (resnet.py:282-290, 315-330)."""
    import hip_ops as H
    r = _rng(seed)
    H2, W2 = (OH - 1) * stride + 1 + (stride > 1), (OW - 1) * stride + 1 + (stride > 1)      # an even-sized input for stride 2
    M = B * OH * OW
    h = _bf(r.standard_normal((M, K1)))
    x = _bf(r.standard_normal((B, H2, W2, K2)))
    w3 = _bf(r.standard_normal((K1, N)) / math.sqrt(K1))
    wd = _bf(r.standard_normal((K2, N)) / math.sqrt(K2))
    bvec = r.standard_normal(N).astype(np.float32)
    xs = x[:, ::stride, ::stride, :][:, :OH, :OW, :].reshape(M, K2)
    ref = h.astype(np.float64) @ w3.astype(np.float64) + xs.astype(np.float64) @ wd.astype(np.float64) + bvec
    if act == "relu":
        ref = np.maximum(ref, 0.0)
    k1p, k2p = -(-K1 // 64) * 64, -(-K2 // 64) * 64
    wt = np.zeros((N, k1p + k2p), np.float32)
    wt[:, :K1] = w3.T
    wt[:, k1p:k1p + K2] = wd.T
    got = H.gemm(H.dev_bf16(h), H.dev_bits(pack.to_bf16_bits(wt)), N, K1, bias=H.dev_f32(bvec), act=act, tile_hint=tile,
                 a2=H.dev_bf16(x.reshape(-1, K2)), a2_geom=None if stride == 1 else (stride, H2, W2, OH, OW))
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


def _dual_conv_case(B, OH, OW, C1, C2, N, stride, seed, tile=0):
    """... with a 3x3 / stride 1 / pad 1 GATHER as the first operand: conv2 of a BASIC block (resnet.py:176-186) + the block's
    1x1 / stride-s shortcut convolution of the block input."""
    import hip_ops as H
    r = _rng(seed)
    H2, W2 = OH * stride, OW * stride
    M = B * OH * OW
    h = _bf(r.standard_normal((B, OH, OW, C1)))
    x = _bf(r.standard_normal((B, H2, W2, C2)))
    kern = _bf(r.standard_normal((3, 3, C1, N)) / math.sqrt(9 * C1))
    wd = _bf(r.standard_normal((C2, N)) / math.sqrt(C2))
    bvec = r.standard_normal(N).astype(np.float32)
    y = O.conv2d(O.zero_pad2d(torch.from_numpy(h), 1), torch.from_numpy(kern), None, stride=1).numpy().reshape(M, N)
    xs = x[:, ::stride, ::stride, :].reshape(M, C2)
    ref = np.maximum(y.astype(np.float64) + xs.astype(np.float64) @ wd.astype(np.float64) + bvec, 0.0)
    wt1, _, K, mode = pack.pack_conv(kern, None, None, C1)
    k2p = -(-C2 // 64) * 64
    wt2 = np.zeros((N, k2p), np.uint16)
    wt2[:, :C2] = pack.to_bf16_bits(wd.T)
    wt = np.concatenate([wt1, wt2], axis=1)
    conv = dict(mode=mode, B=B, H=OH, W=OW, Cin=C1, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, OH=OH, OW=OW)
    got = H.gemm(H.dev_bf16(h.reshape(-1, C1)), H.dev_bits(wt), N, K, conv=conv, bias=H.dev_f32(bvec), act="relu", tile_hint=tile,
                 a2=H.dev_bf16(x.reshape(-1, C2)), a2_geom=None if stride == 1 else (stride, H2, W2, OH, OW))
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


# ResNet-18's strided first blocks (3x3 conv2 gather + the 1x1 / stride-2 shortcut of the block input)
CASES["gemm_dual_conv3x3_resnet18_stage2"] = lambda: _dual_conv_case(3, 28, 28, 128, 64, 128, 2, 160)
CASES["gemm_dual_conv3x3_resnet18_stage4"] = lambda: _dual_conv_case(5, 7, 7, 512, 256, 512, 2, 161)
CASES["gemm_dual_conv3x3_cin72_stride1"] = lambda: _dual_conv_case(2, 9, 11, 72, 40, 96, 1, 162)            # Cin % 64 != 0: per-piece tap division
for _t in (21, 23, 25, 27):
    CASES[f"gemm_dual_conv3x3_tile{_t}"] = lambda t=_t: _dual_conv_case(4, 14, 14, 64, 136, 264, 2, 170 + t, tile=t)


def _dual_window_case(B, OH, OW, K1, K2, N, seed, tile=0):
    """... with a 2 x 2 / stride-2 WINDOW as the second operand (a2_window = 2): ResNet-D's AveragePooling2D(2, 2) + 1x1 convolution
    shortcut (resnet.py:295-312) folded into conv3 -- four taps of the 1x1 kernel / 4."""
    import hip_ops as H
    r = _rng(seed)
    H2, W2 = 2 * OH, 2 * OW
    M = B * OH * OW
    h = _bf(r.standard_normal((M, K1)))
    x = _bf(r.standard_normal((B, H2, W2, K2)))
    w3 = _bf(r.standard_normal((K1, N)) / math.sqrt(K1))
    wd = _bf(r.standard_normal((K2, N)) / math.sqrt(K2))
    bvec = r.standard_normal(N).astype(np.float32)
    pooled = x.reshape(B, OH, 2, OW, 2, K2).astype(np.float64).mean(axis=(2, 4)).reshape(M, K2)
    ref = np.maximum(h.astype(np.float64) @ w3.astype(np.float64) + pooled @ wd.astype(np.float64) + bvec, 0.0)
    k1p, k2p = -(-K1 // 64) * 64, -(-K2 // 64) * 64
    wt = np.zeros((N, k1p + 4 * k2p), np.float32)
    wt[:, :K1] = w3.T
    for t in range(4):
        wt[:, k1p + t * k2p:k1p + t * k2p + K2] = wd.T * 0.25
    got = H.gemm(H.dev_bf16(h), H.dev_bits(pack.to_bf16_bits(wt)), N, K1, bias=H.dev_f32(bvec), act="relu", tile_hint=tile,
                 a2=H.dev_bf16(x.reshape(-1, K2)), a2_geom=(2, H2, W2, OH, OW, 2))
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


# resnet50d's strided first blocks (stage 2: conv3 128 -> 512 + avg-pooled 256 -> 512)
CASES["gemm_dual_window2_resnet50d_stage2"] = lambda: _dual_window_case(3, 28, 28, 128, 256, 512, 180)
CASES["gemm_dual_window2_resnet50d_stage4"] = lambda: _dual_window_case(6, 7, 7, 512, 1024, 2048, 181)
CASES["gemm_dual_window2_ragged"] = lambda: _dual_window_case(2, 5, 9, 72, 40, 96, 182)             # K2 = 40: taps padded to 64 each
for _t in (21, 23, 25, 27):
    CASES[f"gemm_dual_window2_tile{_t}"] = lambda t=_t: _dual_window_case(4, 14, 14, 192, 136, 264, 183 + t, tile=t)


# ResNet-50's three strided first blocks at small batch (stage 2: 128 + 256 -> 512 at 28 x 28; stage 3: 256 + 512 -> 1024; stage 4)
CASES["gemm_dual_resnet_stage2_s2"] = lambda: _dual_case(3, 28, 28, 128, 256, 512, 2, 140)
CASES["gemm_dual_resnet_stage3_s2"] = lambda: _dual_case(5, 14, 14, 256, 512, 1024, 2, 141)
CASES["gemm_dual_resnet_stage4_s2"] = lambda: _dual_case(6, 7, 7, 512, 1024, 2048, 2, 142)
CASES["gemm_dual_stride1_ragged_k"] = lambda: _dual_case(2, 9, 11, 72, 40, 96, 1, 143, act="")        # K1, K2 not whole k-tiles; M = 198
for _t in (21, 22, 23, 24, 25, 26, 27, 29, 30):
    CASES[f"gemm_dual_tile{_t}"] = lambda t=_t: _dual_case(4, 14, 14, 192, 136, 264, 2, 150 + t, tile=t)   # N % 8 == 0 only, two column tiles at 256


CASES["gemm_se_scale_prologue"] = lambda: _se_scale_case(3, 50, 48, 24, 20)
for _t in (1, 21, 22, 23, 24, 25, 26, 27, 29):
    # rows_per_image 144 < tile height: a tile spans 2-3 images; K = 200 is not a whole k-tile / gate piece
    CASES[f"gemm_se_scale_tile{_t:02d}_r144_k200"] = (
        lambda t=_t: _se_scale_case(11, 144, 200, 72, 110 + t, tile=t, residual=True, bias=True))
CASES["gemm_se_scale_t24_r2304_k960"] = lambda: _se_scale_case(5, 2304, 960, 160, 130, tile=24, residual=True)   # several gate pieces per slot
CASES["gemm_se_scale_t23_r9025_k144_multiround"] = lambda: _se_scale_case(9, 9025, 144, 32, 131, tile=23, bias=True)
# (round 6: the gate arrives per k-tile, 1 KiB per four image slots -- csrc/gemm_stream_kernel.h issue_gate)
CASES["gemm_se_scale_t21_r36_k2688"] = lambda: _se_scale_case(40, 36, 2688, 448, 132, tile=21)   # 9 image slots per tile: three gate pieces per k-tile
CASES["gemm_se_scale_t23_r36_k2688"] = lambda: _se_scale_case(40, 36, 2688, 448, 133, tile=23, residual=True)   # 128-row tile, 5 slots: two pieces on four waves
CASES["gemm_se_scale_t21_r4_falls_back"] = lambda: _se_scale_case(70, 4, 200, 72, 134, tile=21, bias=True)     # 65 slots = 17 pieces > 8 waves: register-staged kernel
CASES["gemm_se_scale_t25_r49_k1152"] = lambda: _se_scale_case(37, 49, 1152, 192, 135, tile=25, residual=True)  # 7 x 7 images (EfficientNet-B0's last stage), odd image count


@case("gemm_row_select_lda")
def _():
    """class-token select: A row b = x[b, 0, :] via lda = N_tokens * D (vit.py:462)."""
    import hip_ops as H
    r = _rng(21)
    B, T, D, N = 5, 7, 64, 40
    x = _bf(r.standard_normal((B, T, D)))
    w = _bf(r.standard_normal((D, N)) / 8)
    wt, _ = pack.pack_dense(w, None)
    ref = x[:, 0, :].astype(np.float64) @ w.astype(np.float64)
    got = H.gemm(H.dev_bf16(x), H.dev_bits(wt), N, D, M=B, lda=T * D, out_f32=True)
    H.sync()
    return _err(_cpu(got), ref), TOL_F32


# ---------------------------------------------------------------------------------------------
# convolutions through the GEMM gather modes
# ---------------------------------------------------------------------------------------------
def _conv_case(B, H, W, Cin, Cout, k, stride, padding, *, act="", bn=True, residual=False, seed=0, tile=0, pair=False):
    import hip_ops as Hh
    r = _rng(seed)
    x = _bf(r.standard_normal((B, H, W, Cin)))
    kern = (r.standard_normal((k, k, Cin, Cout)) / math.sqrt(k * k * Cin)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, Cout).astype(np.float32) if bn else None
    shift = r.standard_normal(Cout).astype(np.float32) if bn else None
    cin_stored = pack.pad_channels(Cin) if Cin <= 4 else Cin
    wt, bias, K, mode = pack.pack_conv(kern, scale, shift, cin_stored)
    # reference uses the same folded+rounded weights
    kf = kern * (scale.reshape(1, 1, 1, -1) if bn else 1.0)
    kf = _bf(kf)
    xt = torch.from_numpy(x)
    if padding == "same":
        y = O.conv2d(xt, torch.from_numpy(kf), None, stride=stride, padding="same")
        pt, _ = O.same_pad_amounts(H, k, stride)
        pl, _ = O.same_pad_amounts(W, k, stride)
    else:
        y = O.conv2d(O.zero_pad2d(xt, padding), torch.from_numpy(kf), None, stride=stride)
        pt = pl = padding
    if bn:
        y = y + torch.from_numpy(shift)
    OH, OW = y.shape[1], y.shape[2]
    res = _bf(r.standard_normal((B, OH, OW, Cout))) if residual else None
    if residual:
        y = O.activation(y + torch.from_numpy(res), act)
    else:
        y = O.activation(y, act)
    xd = Hh.dev_bf16(x)
    if pair:
        # RGB stem on the pixel-pair view of the zero-bordered image (what graph.Builder.conv emits)
        assert cin_stored == 4 and stride % 2 == 0
        kwp = (k + 1) // 2 * 2
        wp = max(W + pl, (OW - 1) * stride + kwp)
        wp += wp & 1
        hp = max(H + pt, (OH - 1) * stride + k)
        xd = Hh.cast_input_pad(xd, (pt, hp - H - pt, pl, wp - W - pl))
        conv = dict(mode=1, B=B, H=hp, W=wp // 2, Cin=8, KH=k, KW=kwp // 2, stride=stride, stride_w=stride // 2,
                    pad_t=0, pad_l=0, OH=OH, OW=OW)
    else:
        if cin_stored != Cin:
            xd = Hh.cast_input(xd, cin_stored)
        conv = dict(mode=mode, B=B, H=H, W=W, Cin=cin_stored, KH=k, KW=k, stride=stride, pad_t=pt, pad_l=pl, OH=OH, OW=OW)
    got = Hh.gemm(xd, Hh.dev_bits(wt), Cout, K, bias=None if bias is None else Hh.dev_f32(bias),
                  residual=None if res is None else Hh.dev_bf16(res.reshape(-1, Cout)), act=act,
                  act_after_res=residual, conv=conv, tile_hint=tile)
    Hh.sync()
    return _err(_cpu(got).reshape(B, OH, OW, Cout), y.numpy()), TOL_BF16


CASES["conv3x3_s1_p1_64to64_relu"] = lambda: _conv_case(2, 14, 14, 64, 64, 3, 1, 1, act="relu", seed=30)
CASES["conv3x3_s2_p1_128to128_relu"] = lambda: _conv_case(2, 28, 28, 128, 128, 3, 2, 1, act="relu", seed=31)
CASES["conv3x3_s1_p1_res_relu_after"] = lambda: _conv_case(1, 9, 11, 32, 48, 3, 1, 1, act="relu", residual=True, seed=32)
CASES["conv1x1_s2_256to512"] = lambda: _conv_case(2, 14, 14, 256, 512, 1, 2, 0, seed=33)
CASES["conv7x7_s2_p3_rgb_stem"] = lambda: _conv_case(2, 64, 64, 3, 64, 7, 2, 3, act="relu", seed=34)
CASES["conv16x16_s16_rgb_patch"] = lambda: _conv_case(2, 64, 64, 3, 96, 16, 16, 0, bn=False, seed=35)
CASES["conv4x4_s4_rgb_patch"] = lambda: _conv_case(2, 32, 32, 3, 128, 4, 4, 0, seed=36)
CASES["conv3x3_s2_same_rgb_odd"] = lambda: _conv_case(2, 33, 33, 3, 48, 3, 2, "same", act="swish", seed=37)
CASES["conv3x3_s2_same_even"] = lambda: _conv_case(2, 20, 20, 16, 24, 3, 2, "same", seed=38)
CASES["conv3x3_scalar_cin6"] = lambda: _conv_case(2, 8, 8, 6, 10, 3, 1, 1, act="relu", seed=39)
CASES["conv3x3_scalar_cin2_s2"] = lambda: _conv_case(3, 9, 9, 2, 4, 3, 2, 1, seed=40)
CASES["conv1ch_8x8_patch"] = lambda: _conv_case(2, 32, 32, 1, 4, 8, 8, 0, bn=False, seed=41)
for _t in (1, 2, 3, 4, 5, 6, 11, 12, 13, 14, 15, 16, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30):
    CASES[f"conv3x3_tile{_t:02d}"] = (lambda t=_t: _conv_case(2, 16, 16, 64, 96, 3, 1, 1, act="relu", seed=42, tile=t))
    CASES[f"conv3x3_s2_res_tile{_t:02d}"] = (
        lambda t=_t: _conv_case(3, 15, 13, 40, 72, 3, 2, 1, act="relu", residual=True, seed=43, tile=t))


def _stem_pool_case(B, H, W, seed, cin=3):
    """fused ResNet stem (tfimm_hip_stem_conv_pool) against conv 7x7/2 pad 3 + BN + ReLU + zero-pad 1 + maxpool 3x3/2 of
    the oracle; the convolution output is rounded to bf16 before the pooling, as the unfused engine path stores it"""
    import hip_ops as Hh
    r = _rng(seed)
    k, stride, Cout = 7, 2, 64
    x = _bf(r.standard_normal((B, H, W, cin)))
    kern = (r.standard_normal((k, k, cin, Cout)) / math.sqrt(k * k * cin)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = r.standard_normal(Cout).astype(np.float32)
    wt, bias, K, _ = pack.pack_conv(kern, scale, shift, 4)
    kf = _bf(kern * scale.reshape(1, 1, 1, -1))
    y = O.conv2d(O.zero_pad2d(torch.from_numpy(x), 3), torch.from_numpy(kf), None, stride=stride) + torch.from_numpy(shift)
    y = torch.from_numpy(_bf(O.activation(y, "relu").numpy()))
    OH, OW = y.shape[1], y.shape[2]
    ref = O.max_pool2d(O.zero_pad2d(y, 1), 3, 2).numpy()
    wp = max(W + 3, (OW - 1) * stride + 8)
    wp += wp & 1
    hp = max(H + 3, (OH - 1) * stride + k)
    wd, bd = Hh.dev_bits(wt), Hh.dev_f32(bias)
    xp = Hh.cast_input_pad(Hh.dev_bf16(x), (3, hp - H - 3, 3, wp - W - 3))
    got = Hh.stem_conv_pool(xp, wd, bd, B, hp, wp // 2, OH, OW)
    Hh.sync()
    err = _err(_cpu(got), ref)
    if cin == 3:
        # the same convolution reading the caller's own image (bf16, float32): border, 4th channel and rounding are
        # applied while the LDS ring is filled -- bit-identical to the padded-copy path
        for xin in (Hh.dev_bf16(x), torch.from_numpy(x).to(Hh.DEV)):
            raw = Hh.stem_conv_pool(xin, wd, bd, B, hp, wp // 2, OH, OW, raw=(H, W, 3, 3))
            Hh.sync()
            if not torch.equal(raw, got):
                return float("inf"), TOL_BF16
    return err, TOL_BF16


CASES["stem_pool_224_b3"] = lambda: _stem_pool_case(3, 224, 224, 150)            # bands > 1 (few images)
CASES["stem_pool_160x128"] = lambda: _stem_pool_case(2, 160, 128, 151)
CASES["stem_pool_odd_70x54"] = lambda: _stem_pool_case(2, 70, 54, 152)           # OH = 35 (ragged last step), OW = 27
CASES["stem_pool_tiny_9x11"] = lambda: _stem_pool_case(1, 9, 11, 153)
CASES["stem_pool_many_images"] = lambda: _stem_pool_case(300, 32, 32, 154)       # more items than workgroups
CASES["stem_pool_gray"] = lambda: _stem_pool_case(2, 64, 48, 155, cin=1)
CASES["pair_conv7x7_s2_p3_rgb_stem"] = lambda: _conv_case(2, 64, 64, 3, 64, 7, 2, 3, act="relu", seed=134, pair=True)
CASES["pair_conv7x7_s2_p3_rgb_stem_big"] = lambda: _conv_case(3, 224, 224, 3, 64, 7, 2, 3, act="relu", seed=139, pair=True)
CASES["pair_conv16x16_s16_rgb_patch"] = lambda: _conv_case(2, 64, 64, 3, 96, 16, 16, 0, bn=False, seed=135, pair=True)
CASES["pair_conv4x4_s4_rgb_patch"] = lambda: _conv_case(2, 32, 32, 3, 128, 4, 4, 0, seed=136, pair=True)
CASES["pair_conv3x3_s2_same_rgb_odd"] = lambda: _conv_case(2, 33, 33, 3, 48, 3, 2, "same", act="swish", seed=137, pair=True)
CASES["pair_conv3x3_s2_same_rgb_380"] = lambda: _conv_case(1, 380, 380, 3, 48, 3, 2, "same", act="swish", seed=140, pair=True)
CASES["pair_conv1ch_8x8_patch"] = lambda: _conv_case(2, 32, 32, 1, 8, 8, 8, 0, bn=False, seed=138, pair=True)
for _t in (21, 24, 28, 30):
    CASES[f"conv3x3_stream_multiround_tile{_t:02d}"] = (
        lambda t=_t: _conv_case(32, 56, 56, 64, 64, 3, 1, 1, act="relu", residual=True, seed=70 + t, tile=t))
# 256x256 persistent tile: deep K (many k-tiles per tile, several tiles per workgroup), K tail, two k-tiles
CASES["gemm_t21_deepk_3000x1600x520"] = lambda: _gemm_case(3000, 1600, 520, act="gelu", residual=True, tile=28, seed=91)
CASES["gemm_t21_two_ktiles_70000x128x512"] = lambda: _gemm_case(70000, 128, 512, tile=28, seed=92)
CASES["gemm_t21_ktail_5000x200x256"] = lambda: _gemm_case(5000, 200, 256, act="relu", tile=28, seed=93)
CASES["gemm_t21_resmod_remap_19600x192x768"] = lambda: _gemm_case(19600, 192, 768, residual=True, res_mod=196,
                                                                 remap=(196, 197, 1), tile=21, seed=96)
CASES["conv3x3_t21_cin128_multiround"] = lambda: _conv_case(24, 28, 28, 128, 256, 3, 1, 1, act="relu", residual=True, seed=94, tile=21)
CASES["conv3x3_t21_cin40_s2"] = lambda: _conv_case(16, 31, 29, 40, 264, 3, 2, 1, act="relu", seed=95, tile=21)
CASES["gemm_t29_n24_res_70000x48"] = lambda: _gemm_case(70000, 48, 24, residual=True, tile=29, seed=105)   # narrow-output tile
CASES["gemm_t24_n144_swish_ragged_tiles"] = lambda: _gemm_case(30000, 24, 144, act="swish", tile=24, seed=106)  # last column tile: one wave fully out of range
for _t in (21, 28, 30):
    CASES[f"gemm_t{_t}_resmod_remap_19600x200x768"] = (
        lambda t=_t: _gemm_case(19600, 200, 768, residual=True, res_mod=196, remap=(196, 197, 1), tile=t, seed=97))
    CASES[f"conv3x3_t{_t}_cin40_s2_ragged"] = (
        lambda t=_t: _conv_case(16, 31, 29, 40, 264, 3, 2, 1, act="relu", residual=True, seed=98, tile=t))
    CASES[f"gemm_t{_t}_k32_single_ktile"] = (lambda t=_t: _gemm_case(9000, 32, 520, act="relu", tile=t, seed=99))
# 256x128 two-workgroups-per-CU tile (hint 30): deep K, K tail, two k-tiles, tap stepping at Cin % 32 == 0 and the divide path
CASES["gemm_t30_deepk_3000x1600x520"] = lambda: _gemm_case(3000, 1600, 520, act="gelu", residual=True, tile=30, seed=191)
CASES["gemm_t30_two_ktiles_70000x64x512"] = lambda: _gemm_case(70000, 64, 512, tile=30, seed=192)
CASES["gemm_t30_ktail_5000x200x256"] = lambda: _gemm_case(5000, 200, 256, act="relu", tile=30, seed=193)
CASES["gemm_t30_vit_fc2_20000x3072x768_res"] = lambda: _gemm_case(20000, 3072, 768, residual=True, tile=30, seed=194)
CASES["gemm_t30_relu_after_res_90000x128x512"] = lambda: _gemm_case(90000, 128, 512, act="relu", residual=True, act_after_res=True, tile=30, seed=195)
CASES["conv3x3_t30_cin128_multiround"] = lambda: _conv_case(24, 28, 28, 128, 256, 3, 1, 1, act="relu", residual=True, seed=196, tile=30)
CASES["conv3x3_t30_cin96_s2"] = lambda: _conv_case(16, 31, 29, 96, 264, 3, 2, 1, act="relu", seed=197, tile=30)
CASES["conv1x1_t30_s2_cin64"] = lambda: _conv_case(8, 28, 28, 64, 128, 1, 2, 0, seed=198, tile=30)
CASES["conv3x3_cin128_tapstep"] = lambda: _conv_case(4, 14, 14, 128, 96, 3, 1, 1, act="relu", seed=75)
CASES["conv3x3_s2_cin192_tapstep"] = lambda: _conv_case(3, 15, 15, 192, 64, 3, 2, 1, seed=76)
CASES["conv1x1_s2_cin64_stream"] = lambda: _conv_case(2, 28, 28, 64, 128, 1, 2, 0, seed=77, tile=23)


# ---------------------------------------------------------------------------------------------
# row ops
# ---------------------------------------------------------------------------------------------
def _ln_case(rows, d, eps, seed):
    import hip_ops as H
    r = _rng(seed)
    x = _bf(r.standard_normal((rows, d)) * 2 + 0.5)
    g = r.uniform(0.5, 1.5, d).astype(np.float32)
    b = r.standard_normal(d).astype(np.float32)
    ref = O.layer_norm(torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(b), eps).numpy()
    got = H.layernorm(H.dev_bf16(x), H.dev_f32(g), H.dev_f32(b), eps)
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


CASES["layernorm_768"] = lambda: _ln_case(197 * 3, 768, 1e-6, 50)
CASES["layernorm_192"] = lambda: _ln_case(50, 192, 1e-6, 51)
CASES["layernorm_384_rows_tail"] = lambda: _ln_case(1001, 384, 1e-6, 62)            # 16 lanes x 3 chunks, 4 rows per wave
CASES["layernorm_1024"] = lambda: _ln_case(33, 1024, 1e-5, 52)
CASES["layernorm_2048"] = lambda: _ln_case(9, 2048, 1e-5, 53)
CASES["layernorm_4096"] = lambda: _ln_case(5, 4096, 1e-5, 54)
CASES["layernorm_narrow_128_rows_tail"] = lambda: _ln_case(1003, 128, 1e-5, 57)   # 4 rows per wave, 1003 % 4 != 0
CASES["layernorm_narrow_96"] = lambda: _ln_case(61, 96, 1e-6, 58)                 # 12 of 16 lanes per row
CASES["layernorm_narrow_256"] = lambda: _ln_case(77, 256, 1e-5, 59)               # 2 rows per wave
CASES["layernorm_narrow_64"] = lambda: _ln_case(131, 64, 1e-6, 60)                # 8 rows per wave
CASES["layernorm_narrow_16"] = lambda: _ln_case(19, 16, 1e-6, 61)                 # 2 of 8 lanes per row
CASES["layernorm_generic_d4"] = lambda: _ln_case(34, 4, 1e-6, 55)
CASES["layernorm_generic_d100"] = lambda: _ln_case(7, 100, 1e-5, 56)


@case("layernorm_strided_rows")
def _():
    import hip_ops as H
    r = _rng(57)
    B, T, D = 6, 5, 64
    x = _bf(r.standard_normal((B, T, D)))
    g = r.uniform(0.5, 1.5, D).astype(np.float32)
    b = r.standard_normal(D).astype(np.float32)
    ref = O.layer_norm(torch.from_numpy(x[:, 0]), torch.from_numpy(g), torch.from_numpy(b), 1e-6).numpy()
    got = H.layernorm(H.dev_bf16(x), H.dev_f32(g), H.dev_f32(b), 1e-6, rows=B, d=D, xs=T * D, ys=D)
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


def _attn_ref(qkv, B, N, heads, hd, scale):
    q = qkv.reshape(B, N, 3, heads, hd).transpose(2, 0, 3, 1, 4).astype(np.float64)
    s = scale * (q[0] @ q[1].transpose(0, 1, 3, 2))
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    o = p @ q[2]
    return o.transpose(0, 2, 1, 3).reshape(B * N, heads * hd)


def _attn_case(B, N, heads, hd, seed, spike=False):
    import hip_ops as H
    r = _rng(seed)
    qkv = r.standard_normal((B * N, 3 * heads * hd))
    if spike:  # force a late running-max jump in the online softmax (rescale branch)
        qkv[N - 1, heads * hd: heads * hd + hd] = 6.0 * np.sign(qkv[0, :hd])
    qkv = _bf(qkv)
    scale = hd ** -0.5
    ref = _attn_ref(qkv, B, N, heads, hd, scale)
    got = H.attention(H.dev_bf16(qkv), B, N, heads, hd, scale)
    H.sync()
    return _err(_cpu(got), ref), 1.5e-2   # P is rounded to bf16 before P.V


CASES["attn_vit_197_h3_hd64"] = lambda: _attn_case(2, 197, 3, 64, 60)
CASES["attn_vit_197_spike"] = lambda: _attn_case(1, 197, 2, 64, 61, spike=True)
CASES["attn_64_exact_block"] = lambda: _attn_case(2, 64, 2, 64, 62)
CASES["attn_65_tail1"] = lambda: _attn_case(1, 65, 1, 64, 63)
CASES["attn_577_hd64"] = lambda: _attn_case(1, 577, 2, 64, 64)
CASES["attn_mini_17_hd2"] = lambda: _attn_case(3, 17, 2, 2, 65)
CASES["attn_50_hd32"] = lambda: _attn_case(2, 50, 4, 32, 66)
CASES["attn_hd48"] = lambda: _attn_case(2, 33, 2, 48, 67)
# head dims above 64 (vit_huge_patch14_224_in21k: 1280 / 16 = 80): three / four k-steps over the head dimension
CASES["attn_257_hd80_vit_huge"] = lambda: _attn_case(2, 257, 16, 80, 270)
CASES["attn_70_hd96"] = lambda: _attn_case(2, 70, 3, 96, 271)
CASES["attn_197_hd128"] = lambda: _attn_case(1, 197, 2, 128, 272)
CASES["attn_33_hd72_spike"] = lambda: _attn_case(2, 33, 2, 72, 273, spike=True)


def _tha_ref(qkv, B, N, heads, hd, scale, wl, bl, ww, bw):
    """TalkingHeadAttention.call between qkv and proj (reference cait.py:236-256), float64."""
    q = qkv.reshape(B, N, 3, heads, hd).transpose(2, 0, 3, 1, 4).astype(np.float64)
    s = (scale * q[0]) @ q[1].transpose(0, 1, 3, 2)                  # (B, H, N, N)
    s = s.transpose(0, 2, 3, 1) @ wl.astype(np.float64) + bl         # proj_l over the head axis
    s = s.transpose(0, 3, 1, 2)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    p = p.transpose(0, 2, 3, 1) @ ww.astype(np.float64) + bw         # proj_w
    p = p.transpose(0, 3, 1, 2)
    o = p @ q[2]
    return o.transpose(0, 2, 1, 3).reshape(B * N, heads * hd)


def _tha_case(B, N, heads, hd, seed, use_dev=True):
    import hip_ops as H
    r = _rng(seed)
    qkv = _bf(r.standard_normal((B * N, 3 * heads * hd)))
    wl = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
    ww = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
    bl = (0.3 * r.standard_normal(heads)).astype(np.float32)
    bw = (0.02 * r.standard_normal(heads)).astype(np.float32)
    scale = hd ** -0.5
    ref = _tha_ref(qkv, B, N, heads, hd, scale, wl, bl, ww, bw)
    got = H.talking_heads_attention(H.dev_bf16(qkv), B, N, heads, hd, scale, wl, bl, ww, bw, use_dev=use_dev)
    H.sync()
    return _err(_cpu(got), ref), 1.5e-2   # mixed probabilities are rounded to bf16 before P.V


CASES["tha_196_h4_hd48"] = lambda: _tha_case(2, 196, 4, 48, 160)
CASES["tha_50_h6_hd48_tail"] = lambda: _tha_case(3, 50, 6, 48, 161)
CASES["tha_64_h8_hd48_exact"] = lambda: _tha_case(1, 64, 8, 48, 162)
CASES["tha_100_h16_hd48_two_groups"] = lambda: _tha_case(1, 100, 16, 48, 163)
CASES["tha_33_h2_hd32"] = lambda: _tha_case(2, 33, 2, 32, 164)
CASES["tha_17_h3_hd32"] = lambda: _tha_case(2, 17, 3, 32, 165)
CASES["tha_577_h4_hd48_long"] = lambda: _tha_case(1, 577, 4, 48, 166)
CASES["tha_9_h1_hd32"] = lambda: _tha_case(2, 9, 1, 32, 167)
CASES["tha_196_h4_hd48_weights_by_value"] = lambda: _tha_case(2, 196, 4, 48, 160, use_dev=False)   # no device copy: argument segment
CASES["tha_100_h16_hd48_weights_by_value"] = lambda: _tha_case(1, 100, 16, 48, 163, use_dev=False)
CASES["tha_generic_16_h2_hd2"] = lambda: _tha_case(3, 16, 2, 2, 168)       # the reference's mini: catch-all kernel
CASES["tha_generic_40_h5_hd24"] = lambda: _tha_case(2, 40, 5, 24, 169)


def _class_attn_case(B, N, heads, hd, seed):
    import hip_ops as H
    r = _rng(seed)
    D = heads * hd
    q = _bf(r.standard_normal((B, D)) * hd ** -0.5)                   # already scaled, as the lowering hands it over
    kv = _bf(r.standard_normal((B * N, 2 * D)))
    qh = q.reshape(B, heads, 1, hd).astype(np.float64)
    k = kv[:, :D].reshape(B, N, heads, hd).transpose(0, 2, 1, 3).astype(np.float64)
    v = kv[:, D:].reshape(B, N, heads, hd).transpose(0, 2, 1, 3).astype(np.float64)
    s = qh @ k.transpose(0, 1, 3, 2)                                  # (B, H, 1, N)   cait.py:137
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, D)                 # cait.py:141-143
    got = H.class_attention(H.dev_bf16(q), H.dev_bf16(kv), B, N, heads, hd)
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


CASES["class_attn_197_h4_hd48"] = lambda: _class_attn_case(3, 197, 4, 48, 170)
CASES["class_attn_785_h16_hd48"] = lambda: _class_attn_case(2, 785, 16, 48, 171)
CASES["class_attn_10_h2_hd32"] = lambda: _class_attn_case(2, 10, 2, 32, 172)


def _copy_rows_case():
    import hip_ops as H
    r = _rng(173)
    src = _bf(r.standard_normal((3, 20, 48)))
    dst0 = _bf(r.standard_normal((3, 22, 48)))
    dst = H.dev_bf16(dst0.copy())
    H.copy_rows(H.dev_bf16(src), dst, 1)
    H.sync()
    want = dst0.copy()
    want[:, 1:21] = src
    return float(np.abs(_cpu(dst) - want).max()), 0.0


CASES["copy_rows_concat"] = _copy_rows_case


def _swin_ref(x_qkv, B, Hr, Wr, heads, hd, ws, shift, table):
    """Literal restatement of swin.py:287-318 + WindowAttention.call on a packed qkv tensor."""
    C = heads * hd
    n = ws * ws
    t = torch.from_numpy(x_qkv).reshape(B, Hr, Wr, 3 * C)
    t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    t = t.reshape(B, Hr // ws, ws, Wr // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, n, 3 * C)
    qkv = t.reshape(-1, n, 3, heads, hd).permute(2, 0, 3, 1, 4).double()
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-1, -2)
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    index = rel.sum(-1)
    bias = torch.from_numpy(table[index.reshape(-1)].reshape(n, n, heads)).permute(2, 0, 1).double()
    attn = attn + bias.unsqueeze(0)
    if shift > 0:
        img = np.zeros((1, Hr, Wr, 1))
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.reshape(1, Hr // ws, ws, Wr // ws, ws, 1).transpose(0, 1, 3, 2, 4, 5).reshape(-1, n)
        mask = mw[:, None, :] - mw[:, :, None]
        mask = np.where(mask != 0, -100.0, 0.0)
        nw = mask.shape[0]
        attn = attn.reshape(-1, nw, heads, n, n) + torch.from_numpy(mask)[None, :, None]
        attn = attn.reshape(-1, heads, n, n)
    attn = torch.softmax(attn, -1)
    o = (attn @ v).permute(0, 2, 1, 3).reshape(-1, ws, ws, C)
    o = o.reshape(B, Hr // ws, Wr // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hr, Wr, C)
    o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(B * Hr * Wr, C).numpy(), index


def _swin_case(B, Hr, Wr, heads, hd, ws, shift, seed, tiles=False):
    import hip_ops as H
    r = _rng(seed)
    C = heads * hd
    qkv = _bf(r.standard_normal((B * Hr * Wr, 3 * C)))
    table = r.standard_normal(((2 * ws - 1) ** 2, heads)).astype(np.float32)
    ref, index = _swin_ref(qkv, B, Hr, Wr, heads, hd, ws, shift, table)
    n = ws * ws
    bias = table[index.reshape(-1)].reshape(n, n, heads).transpose(2, 0, 1).copy()
    # tiles: bias + shift mask pre-combined on the host per window kind (what the engine passes)
    bl2 = H.dev_f32(pack.swin_bias_tiles(bias, ws, shift)) if tiles else None
    got = H.attention(H.dev_bf16(qkv), B, Hr * Wr, heads, hd, hd ** -0.5, window=ws, shift=shift, res=(Hr, Wr),
                      rel_bias=H.dev_f32(bias), bias_log2=bl2)
    H.sync()
    return _err(_cpu(got), ref), 1.5e-2


CASES["swin_w7_noshift_14x14_h4_hd32"] = lambda: _swin_case(2, 14, 14, 4, 32, 7, 0, 70)
CASES["swin_w7_shift3_14x14_h4_hd32"] = lambda: _swin_case(2, 14, 14, 4, 32, 7, 3, 71)
CASES["swin_w7_shift3_28x14_rect"] = lambda: _swin_case(1, 28, 14, 2, 32, 7, 3, 72)
CASES["swin_w4_shift2_8x8_hd4"] = lambda: _swin_case(2, 8, 8, 1, 4, 4, 2, 73)
CASES["swin_w12_shift6_24x24"] = lambda: _swin_case(1, 24, 24, 2, 32, 12, 6, 74)
CASES["swin_w7_single_window"] = lambda: _swin_case(3, 7, 7, 2, 32, 7, 0, 75)
CASES["swin_tiles_w7_noshift_14x14"] = lambda: _swin_case(2, 14, 14, 4, 32, 7, 0, 76, tiles=True)
CASES["swin_tiles_w7_shift3_14x14"] = lambda: _swin_case(2, 14, 14, 4, 32, 7, 3, 77, tiles=True)
CASES["swin_tiles_w7_shift3_28x14_rect"] = lambda: _swin_case(1, 28, 14, 2, 32, 7, 3, 78, tiles=True)
CASES["swin_tiles_w7_shift3_21x35"] = lambda: _swin_case(2, 21, 35, 3, 32, 7, 3, 79, tiles=True)
CASES["swin_tiles_w4_shift2_8x8_hd4"] = lambda: _swin_case(2, 8, 8, 1, 4, 4, 2, 80, tiles=True)
CASES["swin_tiles_w12_shift6_24x24"] = lambda: _swin_case(1, 24, 24, 2, 32, 12, 6, 81, tiles=True)
CASES["swin_tiles_w7_shift3_single_row"] = lambda: _swin_case(2, 7, 21, 2, 32, 7, 3, 82, tiles=True)
# (round 6: attn_window_persist_kernel -- chunks of windows of one mask kind, bias in registers)
CASES["swin_tiles_w7_shift3_56x56_b5_chunks"] = lambda: _swin_case(5, 56, 56, 4, 32, 7, 3, 83, tiles=True)     # 320 windows x 2 head pairs: chunks of several windows, all four kinds
CASES["swin_tiles_w7_noshift_56x56_b5_chunks"] = lambda: _swin_case(5, 56, 56, 4, 32, 7, 0, 84, tiles=True)
CASES["swin_tiles_w7_shift3_3heads_odd"] = lambda: _swin_case(3, 28, 28, 3, 32, 7, 3, 85, tiles=True)           # Swin-T's stage 1: odd head count -> one head per workgroup
CASES["swin_tiles_w7_shift3_single_col"] = lambda: _swin_case(2, 21, 7, 2, 32, 7, 3, 86, tiles=True)
# persistent global-attention kernel (>= 2048 (image, head) items, 129..256 tokens, head dim 64): several items per
# workgroup, the last round ragged; 197 / 256 / 129 tokens
CASES["attn_stream_197_hd64"] = lambda: _attn_case(171, 197, 12, 64, 160)
CASES["attn_stream_197_spike"] = lambda: _attn_case(342, 197, 6, 64, 161, spike=True)
CASES["attn_stream_256_hd64"] = lambda: _attn_case(129, 256, 16, 64, 162)
CASES["attn_stream_129_hd64"] = lambda: _attn_case(257, 129, 8, 64, 163)
CASES["attn_stream_200_hd32_stays_resident"] = lambda: _attn_case(64, 200, 17, 32, 164)
CASES["attn_256_exact"] = lambda: _attn_case(1, 256, 2, 64, 68)
CASES["attn_130_hd32"] = lambda: _attn_case(2, 130, 2, 32, 69)


@case("maxpool_3x3_s2_p1")
def _():
    import hip_ops as H
    r = _rng(80)
    x = _bf(np.maximum(r.standard_normal((2, 15, 15, 64)), 0))
    ref = O.max_pool2d(O.zero_pad2d(torch.from_numpy(x), 1), 3, 2).numpy()
    got = H.maxpool(H.dev_bf16(x), 3, 2, 1)
    H.sync()
    e1 = _err(_cpu(got), ref)
    x = _bf(r.standard_normal((2, 9, 9, 6)))   # generic path; NEGATIVE values: zero padding wins at borders
    ref = O.max_pool2d(O.zero_pad2d(torch.from_numpy(x), 1), 3, 2).numpy()
    got = H.maxpool(H.dev_bf16(x), 3, 2, 1)
    H.sync()
    return max(e1, _err(_cpu(got), ref)), 1e-6


@case("mean_rows")
def _():
    import hip_ops as H
    r = _rng(81)
    x = _bf(r.standard_normal((3, 49, 200)))
    got = H.mean_rows(H.dev_bf16(x), out_f32=True)
    got2 = H.mean_rows(H.dev_bf16(x), out_f32=False)
    H.sync()
    ref = x.mean(1)
    return max(_err(_cpu(got), ref), _err(_cpu(got2), ref) / 10), 1e-3


@case("mean_rows_wide_and_odd")
def _():
    """vector path over several 512-channel groups (2048, 520) and the scalar path (channel count not a multiple of 8)"""
    import hip_ops as H
    r = _rng(84)
    worst = 0.0
    for shape in [(5, 49, 2048), (3, 7, 520), (2, 144, 1792), (4, 10, 100), (2, 1, 8)]:
        x = _bf(r.standard_normal(shape))
        got = H.mean_rows(H.dev_bf16(x), out_f32=True)
        got2 = H.mean_rows(H.dev_bf16(x), out_f32=False)
        H.sync()
        ref = x.mean(1)
        worst = max(worst, _err(_cpu(got), ref), _err(_cpu(got2), ref) / 10)
    return worst, 1e-3


@case("bcast_rows")
def _():
    import hip_ops as H
    r = _rng(82)
    B, T, D, n = 4, 6, 40, 2
    src = _bf(r.standard_normal((n, D)))
    dst0 = _bf(r.standard_normal((B, T, D)))
    dst = H.dev_bf16(dst0)
    H.bcast_rows(H.dev_bf16(src), dst, B, n, D, T)
    H.sync()
    ref = dst0.copy()
    ref[:, :n] = src
    return _err(_cpu(dst), ref), 1e-6


@case("cast_input_rgb_and_generic")
def _():
    import hip_ops as H
    r = _rng(83)
    x = r.standard_normal((2, 5, 7, 3)).astype(np.float32)
    got = H.cast_input(torch.from_numpy(x).to(H.DEV), 4)
    got5 = H.cast_input(torch.from_numpy(r.standard_normal((2, 3, 3, 5)).astype(np.float32)).to(H.DEV), 8)
    gotb = H.cast_input(H.dev_bf16(x), 4)
    H.sync()
    ref = np.concatenate([_bf(x), np.zeros((2, 5, 7, 1), np.float32)], -1)
    ok5 = float(np.abs(_cpu(got5)[..., 5:]).max())
    return max(_err(_cpu(got), ref), _err(_cpu(gotb), ref), ok5), 1e-6


@case("cast_input_bf16_rgb_four_pixels_per_thread")
def _():
    """bf16 RGB images whose width is a multiple of four take the 8-byte-load kernels: flat (3 -> 4 channels) and zero-bordered
    with asymmetric pads; bit exact, every border pixel zero, nothing written twice with a different value"""
    import hip_ops as H
    r = _rng(84)
    worst = 0.0
    for (B, Hh, Ww, pad) in ((2, 5, 8, (3, 2, 3, 3)), (3, 7, 12, (0, 1, 2, 0)), (1, 1, 4, (1, 1, 1, 1)), (2, 6, 20, (3, 3, 3, 2))):
        x = _bf(r.standard_normal((B, Hh, Ww, 3)))
        ref4 = np.concatenate([x, np.zeros((B, Hh, Ww, 1), np.float32)], -1)
        flat = H.cast_input(H.dev_bf16(x), 4)
        padded = H.cast_input_pad(H.dev_bf16(x), pad)
        H.sync()
        pt, pb, pl, pr = pad
        refp = np.zeros((B, Hh + pt + pb, Ww + pl + pr, 4), np.float32)
        refp[:, pt:pt + Hh, pl:pl + Ww] = ref4
        worst = max(worst, float(np.abs(_cpu(flat) - ref4).max()), float(np.abs(_cpu(padded) - refp).max()))
    return worst, 0.0


def _preprocess_ref(u8, mean, std):
    """create_preprocessing in float32 on the host (models/factory.py:165-167), then the bf16 rounding of cast_input."""
    x = u8.astype(np.float32) / np.float32(255.0)
    return _bf((x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32))


@case("preprocess_input_u8")
def _():
    """uint8 -> normalised bf16: RGB fast path (pixel counts with every remainder mod 4), generic channel counts, and
    the zero-bordered variant; bit-exact against the host formula (every uint8 value occurs)."""
    import hip_ops as H
    r = _rng(89)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    worst = 0.0
    for shape in [(2, 16, 16, 3), (1, 5, 7, 3), (3, 3, 3, 3), (1, 1, 2, 3)]:
        u = r.integers(0, 256, shape, dtype=np.uint8)
        u.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)[: u.size]
        got = _cpu(H.preprocess_input(torch.from_numpy(u).to(H.DEV), 4, mean, std))
        ref = np.concatenate([_preprocess_ref(u, mean, std), np.zeros(shape[:3] + (1,), np.float32)], -1)
        worst = max(worst, float(np.abs(got - ref).max()))
    for cin, cout in [(1, 8), (5, 8), (8, 8), (3, 8)]:
        m5, s5 = tuple(0.1 * (i + 1) for i in range(cin)), tuple(0.2 + 0.05 * i for i in range(cin))
        u = r.integers(0, 256, (2, 4, 5, cin), dtype=np.uint8)
        got = _cpu(H.preprocess_input(torch.from_numpy(u).to(H.DEV), cout, m5, s5))
        ref = np.concatenate([_preprocess_ref(u, m5, s5), np.zeros((2, 4, 5, cout - cin), np.float32)], -1)
        worst = max(worst, float(np.abs(got - ref).max()))
    for cin, pad in [(3, (3, 3, 3, 3)), (3, (0, 1, 0, 1)), (1, (2, 0, 1, 4)), (4, (1, 1, 1, 1))]:
        m4, s4 = (0.5, 0.4, 0.3, 0.2)[:cin], (0.25, 0.5, 0.2, 0.3)[:cin]
        u = r.integers(0, 256, (2, 9, 6, cin), dtype=np.uint8)
        got = _cpu(H.preprocess_input_pad(torch.from_numpy(u).to(H.DEV), pad, m4, s4))
        ref = np.zeros((2, 9 + pad[0] + pad[1], 6 + pad[2] + pad[3], 4), np.float32)
        ref[:, pad[0]:pad[0] + 9, pad[2]:pad[2] + 6, :cin] = _preprocess_ref(u, m4, s4)
        worst = max(worst, float(np.abs(got - ref).max()))
    H.sync()
    return worst, 0.0


def _dw_case(B, H, W, Cc, k, stride, padding, act, seed):
    import hip_ops as Hh
    r = _rng(seed)
    x = _bf(r.standard_normal((B, H, W, Cc)))
    kern = (r.standard_normal((k, k, Cc, 1)) / k).astype(np.float32)
    scale = r.uniform(0.5, 1.5, Cc).astype(np.float32)
    shift = r.standard_normal(Cc).astype(np.float32)
    w, bias = pack.pack_depthwise(kern, scale, shift)
    kf = torch.from_numpy(kern * scale.reshape(1, 1, -1, 1))
    if padding == "same":
        y = O.depthwise_conv2d(torch.from_numpy(x), kf, None, stride, "same")
        pt, _ = O.same_pad_amounts(H, k, stride)
        pl, _ = O.same_pad_amounts(W, k, stride)
    else:
        y = O.depthwise_conv2d(O.zero_pad2d(torch.from_numpy(x), padding), kf, None, stride)
        pt = pl = padding
    y = O.activation(y + torch.from_numpy(shift), act)
    OH, OW = y.shape[1], y.shape[2]
    got, sums = Hh.dwconv(Hh.dev_bf16(x), Hh.dev_f32(w), Hh.dev_f32(bias), k, stride, pt, pl, OH, OW, act=act,
                          want_sums=True)
    Hh.sync()
    e = _err(_cpu(got), y.numpy())
    es = _err(Hh.sums_to_float(sums), _cpu(got).astype(np.float64).sum((1, 2)))   # sums are of the stored bf16 outputs
    return max(e, es / 10), TOL_BF16




def _act_saturation_case(act, seed):
    """Sigmoid-class activations far outside their usual range: pre-activations of +-25 ... +-3e4, where exp overflows to
    inf or underflows to 0, must give the saturated value -- never NaN.  (A variant of the packed epilogue that shared one
    reciprocal among four denominators 1 + 2^z needed a clamp on z for exactly this; it was slower and is gone, the case
    stays.)  Checked element by element (a bf16 rounding of the exact value, 2e-6 absolute, 1e-9 |v| for a clamped tail),
    through the GEMM epilogue (bias carries the values, the product contributes 0) and through the depthwise kernel."""
    import hip_ops as Hh
    r = _rng(seed)
    vals = np.concatenate([np.linspace(-150, 150, 301), [-3e4, -1e3, -88.8, -24.1, -16.7, 16.7, 24.1, 88.8, 1e3, 3e4],
                           r.standard_normal(201) * 3]).astype(np.float32)
    N = vals.size
    M, K = 300, 16
    a = np.zeros((M, K), np.float32)
    wt, _ = pack.pack_dense(_bf(r.standard_normal((K, N))), None)
    got = _cpu(Hh.gemm(Hh.dev_bf16(a), Hh.dev_bits(wt), N, K, bias=Hh.dev_f32(vals), act=act))
    ref = O.activation(torch.from_numpy(vals), act).numpy().astype(np.float64)
    Hh.sync()
    assert np.all(np.isfinite(got)), "non-finite activation output"
    tol = np.abs(ref) * 2.0 ** -8 + 2e-6 + np.abs(vals.astype(np.float64)) * 1e-9      # last term: the clamped tail (s >= 2^-31)
    worst = float(np.max(np.abs(got.astype(np.float64) - ref[None]) / tol[None]))
    # depthwise 3x3 with a centre tap of 0 and the values as the per-channel shift: act(shift) at every pixel
    Cc = (N + 7) // 8 * 8
    sh = np.zeros(Cc, np.float32)
    sh[:N] = vals
    w, bias = pack.pack_depthwise(np.zeros((3, 3, Cc, 1), np.float32), np.ones(Cc, np.float32), sh)
    x = _bf(r.standard_normal((2, 9, 9, Cc)))
    gd = _cpu(Hh.dwconv(Hh.dev_bf16(x), Hh.dev_f32(w), Hh.dev_f32(bias), 3, 1, 1, 1, 9, 9, act=act)[0])
    Hh.sync()
    assert np.all(np.isfinite(gd)), "non-finite activation output (depthwise)"
    refd = O.activation(torch.from_numpy(sh), act).numpy().astype(np.float64)
    told = np.abs(refd) * 2.0 ** -8 + 2e-6 + np.abs(sh.astype(np.float64)) * 1e-9
    worst = max(worst, float(np.max(np.abs(gd.astype(np.float64) - refd) / told)))
    return worst, 1.0


for _a in ("swish", "sigmoid", "tanh"):
    CASES[f"act_saturation_{_a}"] = (lambda a=_a: _act_saturation_case(a, 410))

CASES["dwconv_k3_s1_same_c48"] = lambda: _dw_case(2, 19, 19, 48, 3, 1, "same", "swish", 90)
CASES["dwconv_k3_s2_same_even"] = lambda: _dw_case(2, 20, 20, 144, 3, 2, "same", "swish", 91)
CASES["dwconv_k5_s2_same_odd"] = lambda: _dw_case(2, 15, 15, 32, 5, 2, "same", "swish", 92)
CASES["dwconv_k5_s2_same_even24"] = lambda: _dw_case(1, 24, 24, 16, 5, 2, "same", "swish", 93)
CASES["dwconv_k5_s1_same_c960"] = lambda: _dw_case(3, 24, 24, 960, 5, 1, "same", "swish", 96)
CASES["dwconv_k3_s2_same_95_c192"] = lambda: _dw_case(2, 95, 95, 192, 3, 2, "same", "swish", 97)
CASES["dwconv_k7_p3_c96_linear"] = lambda: _dw_case(2, 14, 14, 96, 7, 1, 3, "", 94)
CASES["dwconv_k7_p3_56_c96_segments"] = lambda: _dw_case(2, 56, 56, 96, 7, 1, 3, "", 98)       # several row segments, 14 strips
CASES["dwconv_k7_p3_7x7_c768"] = lambda: _dw_case(3, 7, 7, 768, 7, 1, 3, "", 99)               # image smaller than the halo, 3 channel tiles
CASES["dwconv_k7_p3_odd_30x23_c10_gelu"] = lambda: _dw_case(2, 30, 23, 10, 7, 1, 3, "gelu", 100)   # 5 channel pairs, ragged strips, activation
CASES["dwconv_k7_p3_c12_sums"] = lambda: _dw_case(2, 9, 9, 16, 7, 1, 3, "swish", 101)          # squeeze requested: strip kernel
CASES["dwconv_k3_s2_p1_odd_c24_segments"] = lambda: _dw_case(2, 61, 45, 24, 3, 2, 1, "relu6", 107)   # symmetric padding, odd sizes
CASES["dwconv_k5_s2_same_95_c192"] = lambda: _dw_case(2, 95, 95, 192, 5, 2, "same", "swish", 108)
CASES["dwconv_k3_p1_generic_c6"] = lambda: _dw_case(2, 8, 8, 6, 3, 1, 1, "relu", 95)


def _grouped_slice_case(B, H, W, C, groups, stride, act, seed, tile=0):
    """Conv2D(3x3, groups) as one tfimm_hip_gemm launch per group on the group's channel slice (pix_pitch, pointer offsets)"""
    import hip_ops as Hh
    r = _rng(seed)
    w = C // groups
    x = _bf(r.standard_normal((B, H, W, C)))
    kern = (r.standard_normal((3, 3, w, C)) / math.sqrt(9 * w)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, C).astype(np.float32)
    shift = r.standard_normal(C).astype(np.float32)
    kf = _bf(kern * scale.reshape(1, 1, 1, -1))
    y = O.conv2d(O.zero_pad2d(torch.from_numpy(x), 1), torch.from_numpy(kf), None, stride=stride, groups=groups)
    y = O.activation(y + torch.from_numpy(shift), act)
    OH, OW = y.shape[1], y.shape[2]
    xd = Hh.dev_bf16(x)
    out = torch.zeros(B * OH * OW, C, dtype=torch.bfloat16, device=Hh.DEV)
    for g in range(groups):
        sl = slice(g * w, (g + 1) * w)
        wt, bias, K, mode = pack.pack_conv(kern[..., sl], scale[sl], shift[sl], w)
        conv = dict(mode=mode, B=B, H=H, W=W, Cin=w, KH=3, KW=3, stride=stride, pad_t=1, pad_l=1, OH=OH, OW=OW, pix_pitch=C)
        Hh.gemm(xd, Hh.dev_bits(wt), w, K, bias=Hh.dev_f32(bias), act=act, conv=conv, out=out, ldc=C, tile_hint=tile,
                a_byte_offset=g * w * 2, out_byte_offset=g * w * 2)
    Hh.sync()
    return _err(_cpu(out).reshape(y.shape), y.numpy()), TOL_BF16


CASES["grouped_slice_2x64_default"] = lambda: _grouped_slice_case(2, 14, 14, 128, 2, 1, "relu", 320)
CASES["grouped_slice_4x96_s2_generic_k"] = lambda: _grouped_slice_case(2, 15, 13, 384, 4, 2, "relu", 321)      # Cin % 64 != 0
CASES["grouped_slice_2x128_dma_family"] = lambda: _grouped_slice_case(3, 9, 9, 256, 2, 1, "", 322, tile=13)
CASES["grouped_slice_2x64_register_family"] = lambda: _grouped_slice_case(2, 10, 10, 128, 2, 1, "relu", 323, tile=3)
CASES["grouped_slice_2x64_deep_ring"] = lambda: _grouped_slice_case(2, 32, 32, 128, 2, 1, "relu", 324, tile=28)
CASES["grouped_slice_3x8_narrow"] = lambda: _grouped_slice_case(2, 7, 7, 24, 3, 1, "relu", 325)


def _row_stats_case(rows, d, seed, offset=0.0):
    import hip_ops as Hh
    r = _rng(seed)
    x = _bf(r.standard_normal((rows, d)) * r.uniform(0.2, 3.0, (rows, 1)) + offset)
    got = _cpu(Hh.row_stats(Hh.dev_bf16(x), 1e-6))
    x64 = x.astype(np.float64)
    ref = np.stack([x64.mean(1), 1.0 / np.sqrt(x64.var(1) + 1e-6)], 1)
    Hh.sync()
    return float(np.abs(got - ref).max() / np.abs(ref).max()), 1e-5


for _d in (96, 128, 192, 256, 384, 512, 768, 1024, 1536):
    CASES[f"row_stats_d{_d}"] = (lambda d: lambda: _row_stats_case(203, d, 330 + d, offset=2.0))(_d)


def _ln_gemm_case(M, K, N, act, seed, tile=0, offset=0.0, bias=True):
    """Dense layer with the LayerNormalization in front of it folded in (ln_stats / ln_c1) against LN -> matmul in fp64.
    ``offset`` shifts the rows' means far from zero: the rank-1 correction then has to cancel a term much larger than the
    result (the three-way bf16 splits carry it)."""
    import hip_ops as Hh
    r = _rng(seed)
    x = _bf(r.standard_normal((M, K)) * r.uniform(0.5, 2.0, (M, 1)) + offset * r.uniform(0.5, 1.5, (M, 1)))
    gam = r.uniform(0.5, 1.5, K).astype(np.float32)
    bet = (0.3 * r.standard_normal(K)).astype(np.float32)
    w = (r.standard_normal((K, N)) / math.sqrt(K)).astype(np.float32)
    b = r.standard_normal(N).astype(np.float32) if bias else np.zeros(N, np.float32)
    eps = 1e-6
    x64 = x.astype(np.float64)
    ln = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + eps) * gam + bet
    y = O.activation(torch.from_numpy((ln @ w.astype(np.float64) + b).astype(np.float32)), act).numpy()
    wf = (w.astype(np.float64) * gam.reshape(K, 1)).astype(np.float32)
    bf = (bet.astype(np.float64) @ w.astype(np.float64) + b).astype(np.float32)
    wt, bvec = pack.pack_dense(wf, bf)
    c1 = pack.pack_ln_c1(wt, N, K)
    if _TIGHT:
        # the arithmetic of the launch itself: exact statistics, the bf16-rounded folded weights it multiplies with, fp64
        # accumulation, ONE rounding (of the stored output)
        wr = pack.bf16_bits_to_f32(wt).astype(np.float64)[:N, :K]                # [N][K], gamma folded
        xh = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + eps)
        y = O.activation(torch.from_numpy(xh @ wr.T + bvec.astype(np.float64)), act).numpy()
    xd = Hh.dev_bf16(x)
    st = Hh.row_stats(xd, eps)
    got = Hh.gemm(xd, Hh.dev_bits(wt), N, K, bias=Hh.dev_f32(bvec), act=act, tile_hint=tile, ln_stats=st, ln_c1=Hh.dev_bits(c1))
    Hh.sync()
    return _err(_cpu(got), y), TOL_BF16


CASES["ln_gemm_768_2304_qkv"] = lambda: _ln_gemm_case(600, 768, 2304, "", 340)
CASES["ln_gemm_768_3072_gelu"] = lambda: _ln_gemm_case(520, 768, 3072, "gelu", 341)
CASES["ln_gemm_mean_offset_20"] = lambda: _ln_gemm_case(300, 512, 512, "", 342, offset=20.0)
CASES["ln_gemm_128_384_tile128"] = lambda: _ln_gemm_case(1000, 128, 384, "", 343, tile=23)
CASES["ln_gemm_ragged_rows_cols"] = lambda: _ln_gemm_case(333, 192, 200, "gelu", 344, offset=3.0)
CASES["ln_gemm_256x128_tile"] = lambda: _ln_gemm_case(777, 256, 768, "", 345, tile=22)
CASES["ln_gemm_256x64_tile"] = lambda: _ln_gemm_case(515, 384, 192, "", 346, tile=24, bias=False)
CASES["ln_gemm_128x256_tile"] = lambda: _ln_gemm_case(400, 1024, 1024, "gelu", 347, tile=26)
for _t in (21, 22, 23, 24, 25, 26, 27, 29, 30):
    CASES[f"ln_gemm_ragged_tile{_t}"] = (lambda t: lambda: _ln_gemm_case(333, 192, 200, "gelu", 350 + t, tile=t, offset=3.0))(_t)
CASES["ln_gemm_one_k_tile"] = lambda: _ln_gemm_case(5000, 32, 96, "", 349, offset=1.0)        # nk = 1: the table DMA has to be waited for explicitly
CASES["ln_gemm_t30_multi_round_gelu"] = lambda: _ln_gemm_case(70000, 768, 384, "gelu", 351, tile=30, offset=1.0)
CASES["ln_gemm_t30_two_ktiles"] = lambda: _ln_gemm_case(70000, 64, 256, "", 352, tile=30, offset=2.0)
CASES["ln_gemm_multi_round"] = lambda: _ln_gemm_case(70000, 128, 256, "", 348, offset=1.0)


def _expand_dw_many_images_case(B, H, W, cin, c, k, stride, seed):
    """many workgroups per CU at once (the batch sizes that are benchmarked): the fused launch must equal the two-launch path
    (tfimm_hip_gemm + tfimm_hip_dwconv) to a bf16 ulp and itself bit for bit -- a missing barrier behind the halo staging
    passed every small-batch case and broke at 64+ images"""
    import hip_ops as Hh
    r = _rng(seed)
    x = _bf(r.standard_normal((B, H, W, cin)))
    k1 = (r.standard_normal((cin, c)) / np.sqrt(cin)).astype(np.float32)
    t1 = (0.5 * r.standard_normal(c)).astype(np.float32)
    kd = (r.standard_normal((k, k, c, 1)) / k).astype(np.float32)
    t2 = (0.5 * r.standard_normal(c)).astype(np.float32)
    cpad = pack.ceil_to(c, 32)
    wd, b2 = pack.pack_depthwise(kd, None, t2)

    def padc(a):
        out = np.zeros(a.shape[:-1] + (cpad,), np.float32)
        out[..., :c] = a
        return out
    OH, OW = -(-H // stride), -(-W // stride)
    pt, _ = O.same_pad_amounts(H, k, stride)
    pl, _ = O.same_pad_amounts(W, k, stride)
    xd = Hh.dev_bf16(x)
    wt, bvec = pack.pack_dense(k1, t1)
    e = Hh.gemm(xd.reshape(-1, cin), Hh.dev_bits(wt), c, cin, bias=Hh.dev_f32(bvec), act="swish")
    ref, ref_s = Hh.dwconv(e.reshape(B, H, W, c), Hh.dev_f32(wd), Hh.dev_f32(b2), k, stride, pt, pl, OH, OW, act="swish", want_sums=True)
    args = (xd, Hh.dev_bits(pack.pack_expand_frag(k1, cpad)), Hh.dev_f32(padc(t1)), Hh.dev_f32(padc(wd)), Hh.dev_f32(padc(b2)),
            c, k, stride, pt, pl, OH, OW)
    got, sums = Hh.expand_dwconv(*args, act="swish", want_sums=True)
    got2, sums2 = Hh.expand_dwconv(*args, act="swish", want_sums=True)
    Hh.sync()
    assert torch.equal(got, got2) and torch.equal(sums, sums2), "two launches of the fused kernel differ"
    return max(_err(_cpu(got), _cpu(ref)), _err(Hh.sums_to_float(sums), Hh.sums_to_float(ref_s))), TOL_BF16


CASES["expand_dw_b96_k3s2_vs_two_launch"] = lambda: _expand_dw_many_images_case(96, 94, 94, 24, 144, 3, 2, 370)
CASES["expand_dw_b96_k3s1_vs_two_launch"] = lambda: _expand_dw_many_images_case(96, 47, 47, 32, 192, 3, 1, 371)
CASES["expand_dw_b96_k5s2_vs_two_launch"] = lambda: _expand_dw_many_images_case(96, 47, 47, 32, 192, 5, 2, 372)


def _stem_dw_case(B, H, W, cin, c, act, seed, padding="same"):
    """stem flavour of tfimm_hip_expand_dwconv: 3x3 / stride 2 convolution of the RGB image + act (rounded to bf16) followed by
    a 3x3 depthwise layer + act, against the same two layers in fp32"""
    import hip_ops as Hh
    r = _rng(seed)
    x = _bf(r.standard_normal((B, H, W, cin)))
    ks = (r.standard_normal((3, 3, cin, c)) / math.sqrt(9 * cin)).astype(np.float32)
    s1, t1 = r.uniform(0.5, 1.5, c).astype(np.float32), (0.5 * r.standard_normal(c)).astype(np.float32)
    kd = (r.standard_normal((3, 3, c, 1)) / 3).astype(np.float32)
    s2, t2 = r.uniform(0.5, 1.5, c).astype(np.float32), (0.5 * r.standard_normal(c)).astype(np.float32)
    cpad = pack.ceil_to(c, 32)
    ksf = ks * s1.reshape(1, 1, 1, c)
    xt = torch.from_numpy(x)
    if padding == "same":
        e = O.conv2d(xt, torch.from_numpy(_bf(ksf)), None, stride=2, padding="same")
        pt, _ = O.same_pad_amounts(H, 3, 2)
        pl, _ = O.same_pad_amounts(W, 3, 2)
    else:
        e = O.conv2d(O.zero_pad2d(xt, padding), torch.from_numpy(_bf(ksf)), None, stride=2)
        pt = pl = padding
    SH, SW = e.shape[1], e.shape[2]
    e = torch.from_numpy(_bf(O.activation(e + torch.from_numpy(t1), act).numpy()))
    kf = torch.from_numpy(kd * s2.reshape(1, 1, -1, 1))
    y = O.activation(O.depthwise_conv2d(e, kf, None, 1, "same") + torch.from_numpy(t2), act)
    wd, b2 = pack.pack_depthwise(kd, s2, t2)

    def padc(a):
        out = np.zeros(a.shape[:-1] + (cpad,), np.float32)
        out[..., :c] = a
        return out
    hp, wp = max(H + pt, (SH - 1) * 2 + 3), max(W + pl, (SW - 1) * 2 + 3)
    img = Hh.cast_input_pad(Hh.dev_bf16(x), (pt, hp - H - pt, pl, wp - W - pl))
    got, sums = Hh.expand_dwconv(img, Hh.dev_bits(pack.pack_stem_frag(ksf, cpad)), Hh.dev_f32(padc(t1)), Hh.dev_f32(padc(wd)),
                                 Hh.dev_f32(padc(b2)), c, 3, 1, 1, 1, SH, SW, act=act, want_sums=True, stem_hw=(SH, SW))
    Hh.sync()
    return max(_err(_cpu(got), y.numpy()), _err(Hh.sums_to_float(sums), _cpu(got).astype(np.float64).sum((1, 2))) / 10), TOL_BF16


CASES["stem_dw_rgb_64_to_32_swish"] = lambda: _stem_dw_case(2, 64, 64, 3, 32, "swish", 360)
CASES["stem_dw_rgb_odd_75x53_c48"] = lambda: _stem_dw_case(2, 75, 53, 3, 48, "swish", 361)              # odd sizes, 1.5 chunks, ragged tiles
CASES["stem_dw_rgb_symmetric_pad_relu6"] = lambda: _stem_dw_case(3, 48, 80, 3, 32, "relu6", 362, padding=1)   # MobileNet-V2 style
CASES["stem_dw_gray_1ch"] = lambda: _stem_dw_case(1, 40, 40, 1, 40, "swish", 363)


def _expand_dw_case(B, H, W, cin, c, k, stride, padding, act, seed, squeeze=True):
    """tfimm_hip_expand_dwconv against 1x1 conv + act (rounded to bf16, as the two-launch path stores it) + depthwise + act"""
    import hip_ops as Hh
    r = _rng(seed)
    x = _bf(r.standard_normal((B, H, W, cin)))
    k1 = (r.standard_normal((cin, c)) / np.sqrt(cin)).astype(np.float32)
    s1, t1 = r.uniform(0.5, 1.5, c).astype(np.float32), (0.5 * r.standard_normal(c)).astype(np.float32)
    kd = (r.standard_normal((k, k, c, 1)) / k).astype(np.float32)
    s2, t2 = r.uniform(0.5, 1.5, c).astype(np.float32), (0.5 * r.standard_normal(c)).astype(np.float32)
    cpad = pack.ceil_to(c, 32)
    w1s = _bf(k1 * s1.reshape(1, c))                        # the kernel multiplies bf16 weights
    frag = pack.pack_expand_frag(k1 * s1.reshape(1, c), cpad)
    wd, b2 = pack.pack_depthwise(kd, s2, t2)

    def padc(a):
        out = np.zeros(a.shape[:-1] + (cpad,), np.float32)
        out[..., :c] = a
        return out
    e = O.activation(torch.from_numpy(x.reshape(-1, cin).astype(np.float64) @ w1s.astype(np.float64)).float()
                     + torch.from_numpy(t1), act)
    e = torch.from_numpy(_bf(e.numpy())).reshape(B, H, W, c)
    kf = torch.from_numpy(kd * s2.reshape(1, 1, -1, 1))
    if padding == "same":
        y = O.depthwise_conv2d(e, kf, None, stride, "same")
        pt, _ = O.same_pad_amounts(H, k, stride)
        pl, _ = O.same_pad_amounts(W, k, stride)
    else:
        y = O.depthwise_conv2d(O.zero_pad2d(e, padding), kf, None, stride)
        pt = pl = padding
    y = O.activation(y + torch.from_numpy(t2), act)
    OH, OW = y.shape[1], y.shape[2]
    got, sums = Hh.expand_dwconv(Hh.dev_bf16(x), Hh.dev_bits(frag), Hh.dev_f32(padc(t1)),
                                 Hh.dev_f32(padc(wd)), Hh.dev_f32(padc(b2)), c, k, stride, pt, pl, OH, OW, act=act,
                                 want_sums=squeeze)
    Hh.sync()
    err = _err(_cpu(got), y.numpy())
    if squeeze:
        err = max(err, _err(Hh.sums_to_float(sums), _cpu(got).astype(np.float64).sum((1, 2))) / 10)      # sums are of the stored bf16 outputs
    return err, TOL_BF16


CASES["expand_dw_k3_s1_24_144_odd"] = lambda: _expand_dw_case(2, 19, 37, 24, 144, 3, 1, "same", "swish", 300)
CASES["expand_dw_k3_s2_even_to_odd"] = lambda: _expand_dw_case(2, 38, 30, 24, 144, 3, 2, "same", "swish", 301)
CASES["expand_dw_k3_s1_32_192_tiles"] = lambda: _expand_dw_case(3, 47, 70, 32, 192, 3, 1, "same", "swish", 302)    # 4 x 3 tiles, ragged edges
CASES["expand_dw_k5_s2_32_192"] = lambda: _expand_dw_case(2, 31, 45, 32, 192, 5, 2, "same", "swish", 303)
CASES["expand_dw_k3_s2_p1_relu6_16_96"] = lambda: _expand_dw_case(2, 28, 28, 16, 96, 3, 2, 1, "relu6", 304, squeeze=False)   # MobileNet-V2 style
CASES["expand_dw_k3_s1_8_40_partial_chunk"] = lambda: _expand_dw_case(2, 9, 9, 8, 40, 3, 1, "same", "swish", 305)    # C = 32 + 8
CASES["expand_dw_k5_s2_tiny_image"] = lambda: _expand_dw_case(1, 5, 4, 24, 48, 5, 2, "same", "swish", 306)      # image smaller than one tile


def _se_case(B, R, Cc, rd, seed):
    import hip_ops as H
    r = _rng(seed)
    x = _bf(r.standard_normal((B, R, Cc)))
    w1 = (r.standard_normal((rd, Cc)) / 12).astype(np.float32)
    b1 = r.standard_normal(rd).astype(np.float32)
    w2 = (r.standard_normal((Cc, rd)) / 2).astype(np.float32)
    b2 = r.standard_normal(Cc).astype(np.float32)
    sums = x.sum(1)
    mean = torch.from_numpy(sums / R)
    hid = O.activation(mean @ torch.from_numpy(w1.T) + torch.from_numpy(b1), "swish")
    gate = torch.sigmoid(hid @ torch.from_numpy(w2.T) + torch.from_numpy(b2)).numpy()
    g = H.se_gate(H.dev_f32(sums), 1.0 / R, H.dev_f32(w1), H.dev_f32(b1), H.dev_f32(np.ascontiguousarray(w2.T)), H.dev_f32(b2), "swish")
    # the same sums as 64-bit fixed point (what the depthwise kernels accumulate): identical gate up to the 2^-20 quantum
    q = torch.from_numpy(np.rint(sums.astype(np.float64) * 2.0 ** 20).astype(np.int64)).to(H.DEV)
    gq = H.se_gate(q, 1.0 / R, H.dev_f32(w1), H.dev_f32(b1), H.dev_f32(np.ascontiguousarray(w2.T)), H.dev_f32(b2), "swish")
    assert float((gq - g).abs().max()) < 1e-5
    res = _bf(r.standard_normal((B, R, Cc)))
    y = H.scale_channels(H.dev_bf16(x), g, H.dev_bf16(res), relu_after=True)
    H.sync()
    ref_y = np.maximum(x * gate[:, None, :] + res, 0)
    return max(_err(_cpu(g), gate), _err(_cpu(y), ref_y)), TOL_BF16


CASES["se_gate_and_scale"] = lambda: _se_case(3, 25, 144, 6, 96)           # 3 images: one partly filled group of 4
CASES["se_gate_and_scale_b9_c1632"] = lambda: _se_case(9, 4, 1632, 68, 103)    # EfficientNet-B4's widest gate, 3 groups
CASES["se_gate_and_scale_b4_c24_rd1"] = lambda: _se_case(4, 9, 24, 1, 104)


def _patch_merge_case(B, Hh, Ww, Cc, seed):
    import hip_ops as H
    r = _rng(seed)
    x = _bf(r.standard_normal((B, Hh * Ww, Cc)) + 0.5)
    g = r.uniform(0.5, 1.5, 4 * Cc).astype(np.float32)
    b = r.standard_normal(4 * Cc).astype(np.float32)
    t = torch.from_numpy(x).reshape(B, Hh, Ww, Cc)
    cat = torch.cat((t[:, 0::2, 0::2], t[:, 1::2, 0::2], t[:, 0::2, 1::2], t[:, 1::2, 1::2]), -1)   # swin.py:353-357
    ref = O.layer_norm(cat.reshape(B, -1, 4 * Cc), torch.from_numpy(g), torch.from_numpy(b), 1e-5).numpy()
    got = H.patch_merge_ln(H.dev_bf16(x), H.dev_f32(g), H.dev_f32(b), Hh, Ww, 1e-5)
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


CASES["patch_merge_ln"] = lambda: _patch_merge_case(2, 6, 8, 16, 97)            # one partial chunk set (64 of 64 lanes: 8 chunks)
CASES["patch_merge_ln_c128_swin_stage1"] = lambda: _patch_merge_case(2, 14, 14, 128, 98)   # 1 chunk per lane
CASES["patch_merge_ln_c192"] = lambda: _patch_merge_case(1, 4, 6, 192, 99)        # 96 chunks: 2 per lane, half masked
CASES["patch_merge_ln_c512"] = lambda: _patch_merge_case(1, 4, 4, 512, 100)       # 4 chunks per lane
CASES["patch_merge_ln_c1024"] = lambda: _patch_merge_case(1, 2, 2, 1024, 101)     # 8 chunks per lane
CASES["patch_merge_ln_c12_scalar"] = lambda: _patch_merge_case(2, 4, 4, 12, 102)  # C % 8 != 0: element-wise kernel


# ---------------------------------------------------------------------------------------------
# feature-path / ResNet-variant kernels (csrc/features.hip)
# ---------------------------------------------------------------------------------------------
def _attn_probs_case(B, n, heads, hd, seed):
    import hip_ops as H
    r = _rng(seed)
    D = heads * hd
    qkv = _bf(r.standard_normal((B, n, 3 * D)))
    scale = hd ** -0.5
    t = torch.from_numpy(qkv).reshape(B, n, 3, heads, hd).permute(2, 0, 3, 1, 4).double()     # vit.py:156-158
    ref = torch.softmax(scale * (t[0] @ t[1].transpose(-1, -2)), -1).numpy()
    got = H.attention_probs(H.dev_bf16(qkv), B, n, heads, hd, scale)
    H.sync()
    return _err(_cpu(got), ref), TOL_F32


CASES["attn_probs_vit_197_h3_hd64"] = lambda: _attn_probs_case(2, 197, 3, 64, 160)
CASES["attn_probs_hd2_n17"] = lambda: _attn_probs_case(3, 17, 2, 2, 161)          # the reference's mini: scalar loads
CASES["attn_probs_n577_hd48"] = lambda: _attn_probs_case(1, 577, 2, 48, 162)      # 10 keys per lane


def _group_norm_case(B, Hh, Ww, Cc, groups, seed, act="", residual=False):
    import hip_ops as H
    r = _rng(seed)
    x = _bf(r.standard_normal((B, Hh, Ww, Cc)) * r.uniform(0.5, 2.0, Cc) + r.standard_normal(Cc))
    g = r.uniform(0.5, 1.5, Cc).astype(np.float32)
    b = r.standard_normal(Cc).astype(np.float32)
    res = _bf(r.standard_normal((B, Hh, Ww, Cc))) if residual else None
    ref = O.group_norm(torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(b), groups, 1e-5)
    if residual:
        ref = O.activation(ref + torch.from_numpy(res), act)
    else:
        ref = O.activation(ref, act)
    got = H.group_norm(H.dev_bf16(x.reshape(B, Hh * Ww, Cc)), H.dev_f32(g), H.dev_f32(b), groups, 1e-5,
                       act="" if residual else act, residual=None if res is None else H.dev_bf16(res.reshape(B, -1, Cc)),
                       act_after=act if residual else "")
    H.sync()
    return _err(_cpu(got).reshape(ref.shape), ref.numpy()), TOL_BF16


CASES["group_norm_c64_g32_56x56_relu"] = lambda: _group_norm_case(3, 56, 56, 64, 32, 163, act="relu")     # group size 2
CASES["group_norm_c256_g32_res_relu"] = lambda: _group_norm_case(2, 14, 14, 256, 32, 164, act="relu", residual=True)
CASES["group_norm_c2048_g32_7x7"] = lambda: _group_norm_case(2, 7, 7, 2048, 32, 165)                      # one row per pass
CASES["group_norm_c96_g32"] = lambda: _group_norm_case(2, 9, 5, 96, 32, 166, act="relu")                  # group size 3: a vector spans groups
CASES["group_norm_c12_g3_scalar"] = lambda: _group_norm_case(2, 5, 7, 12, 3, 167)                         # C % 8 != 0
CASES["group_norm_c2560_two_vector_blocks"] = lambda: _group_norm_case(1, 3, 3, 2560, 32, 168)            # 320 channel vectors > 256 threads


def _blur_case(B, Hh, Ww, Cc, stride, seed):
    import hip_ops as H
    r = _rng(seed)
    x = _bf(r.standard_normal((B, Hh, Ww, Cc)))
    ref = O.blur_pool2d(torch.from_numpy(x), stride).numpy()
    got = H.blur_pool(H.dev_bf16(x), stride)
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


CASES["blur_pool_s2_even_c64"] = lambda: _blur_case(2, 16, 12, 64, 2, 169)
CASES["blur_pool_s2_odd_c24"] = lambda: _blur_case(2, 15, 13, 24, 2, 170)        # reflect at both borders
CASES["blur_pool_s2_c10_scalar"] = lambda: _blur_case(2, 6, 7, 10, 2, 171)
CASES["blur_pool_s1_c8"] = lambda: _blur_case(1, 5, 5, 8, 1, 172)


def _avg_pool_case(B, Hh, Ww, Cc, k, stride, seed):
    import hip_ops as H
    r = _rng(seed)
    x = _bf(r.standard_normal((B, Hh, Ww, Cc)))
    ref = O.avg_pool2d_same(torch.from_numpy(x), k, stride).numpy()
    got = H.avg_pool(H.dev_bf16(x), k, stride)
    H.sync()
    return _err(_cpu(got), ref), TOL_BF16


CASES["avg_pool_2x2_s2_odd_15x13_c32"] = lambda: _avg_pool_case(2, 15, 13, 32, 2, 2, 173)    # clipped last row and column
CASES["avg_pool_2x2_s2_even_c64"] = lambda: _avg_pool_case(2, 8, 6, 64, 2, 2, 174)
CASES["avg_pool_3x3_s2_c12_scalar"] = lambda: _avg_pool_case(2, 7, 9, 12, 3, 2, 175)          # padding on both sides


def _eca_case(B, Cc, k, seed):
    import hip_ops as H
    r = _rng(seed)
    mean = r.standard_normal((B, Cc)).astype(np.float32)
    w = r.standard_normal(k).astype(np.float32)
    pad = (k - 1) // 2
    t = torch.nn.functional.conv1d(torch.nn.functional.pad(torch.from_numpy(mean)[:, None, :], (pad, pad)),
                                   torch.from_numpy(w).reshape(1, 1, -1))[:, 0, :]          # layers/attention.py:123-125
    ref = torch.sigmoid(t).numpy()
    got = H.eca_gate(H.dev_f32(mean * 49), 1.0 / 49, H.dev_f32(w))
    H.sync()
    return _err(_cpu(got), ref), TOL_F32


CASES["eca_gate_c2048_k7"] = lambda: _eca_case(3, 2048, 7, 176)
CASES["eca_gate_c32_k3"] = lambda: _eca_case(2, 32, 3, 177)


# ---------------------------------------------------------------------------------------------
# fused bottleneck tail (csrc/gemm_chain_kernel.h): 3x3 conv + BN + relu -> 1x1 conv + BN + residual + relu
# ---------------------------------------------------------------------------------------------
def _chain_case(B, Hh, Ww, C1, N2, stride, seed, residual=True):
    import hip_ops as H
    r = _rng(seed)
    cin = C1
    x = _bf(r.standard_normal((B, Hh, Ww, cin)))
    k1 = (r.standard_normal((3, 3, cin, C1)) / math.sqrt(9 * cin)).astype(np.float32)
    s1, t1 = r.uniform(0.5, 1.5, C1).astype(np.float32), r.standard_normal(C1).astype(np.float32)
    k2 = (r.standard_normal((1, 1, C1, N2)) / math.sqrt(C1)).astype(np.float32)
    s2, t2 = r.uniform(0.5, 1.5, N2).astype(np.float32), r.standard_normal(N2).astype(np.float32)
    wt1, b1, K1, mode = pack.pack_conv(k1, s1, t1, cin)
    wt2, b2 = pack.pack_dense((k2.reshape(C1, N2) * s2.reshape(1, N2))[pack.chain_k_order(C1)], t2)
    # reference: the two-launch path's arithmetic -- folded weights rounded to bf16, intermediate rounded to bf16 once
    mid = O.conv2d(O.zero_pad2d(torch.from_numpy(x), 1), torch.from_numpy(_bf(k1 * s1.reshape(1, 1, 1, -1))), None, stride=stride)
    mid = torch.from_numpy(_bf(torch.relu(mid + torch.from_numpy(t1)).numpy()))
    OH, OW = mid.shape[1], mid.shape[2]
    y = O.conv2d(mid, torch.from_numpy(_bf(k2 * s2.reshape(1, 1, 1, -1)))) + torch.from_numpy(t2)
    res = _bf(r.standard_normal((B, OH, OW, N2))) if residual else None
    if residual:
        y = y + torch.from_numpy(res)
    y = torch.relu(y).numpy()
    got = H.conv_chain(H.dev_bf16(x), H.dev_bits(wt1), H.dev_f32(b1), H.dev_bits(wt2), H.dev_f32(b2),
                       None if res is None else H.dev_bf16(res.reshape(-1, N2)), KH=3, KW=3, stride=stride, pad=1,
                       OH=OH, OW=OW, C1=C1, N2=N2)
    H.sync()
    return _err(_cpu(got).reshape(B, OH, OW, N2), y), TOL_BF16


CASES["chain_56x56_b3"] = lambda: _chain_case(3, 56, 56, 64, 256, 1, 180)                      # ResNet-50 stage 1: 37 tiles, ragged last
CASES["chain_odd_57x41"] = lambda: _chain_case(2, 57, 41, 64, 256, 1, 181)                     # odd sizes: every border case of the tap mask
CASES["chain_no_residual_tiny"] = lambda: _chain_case(1, 5, 7, 64, 256, 1, 182, residual=False)   # one partial tile, strip mostly out of range
CASES["chain_63_wide"] = lambda: _chain_case(2, 9, 63, 64, 256, 1, 183)                        # the widest row the strip holds
CASES["chain_multiround_b40"] = lambda: _chain_case(40, 56, 56, 64, 256, 1, 185)               # 490 tiles: two per workgroup
CASES["chain_multiround_b130_28x28"] = lambda: _chain_case(130, 28, 28, 64, 256, 1, 186)       # 399 tiles, images smaller than a tile... of 256 pixels
CASES["chain_n512"] = lambda: _chain_case(3, 14, 14, 64, 512, 1, 187)                          # four GEMM-2 steps
CASES["chain_single_image_1x1"] = lambda: _chain_case(3, 1, 1, 64, 256, 1, 188)                # every tap but the centre masked
# C1 = 128 (ResNet stage 2): the input-strip kernel with the 1x1 convolution chained behind it (csrc/conv_strip.hip)
CASES["chain128_28x28_b6"] = lambda: _chain_case(6, 28, 28, 128, 512, 1, 420)                  # 37 tiles, the last one ragged
CASES["chain128_odd_17x23"] = lambda: _chain_case(3, 17, 23, 128, 512, 1, 421)
CASES["chain128_31_wide_n256"] = lambda: _chain_case(2, 9, 31, 128, 256, 1, 422)               # widest row; four GEMM-2 slices
CASES["chain128_no_residual_tiny"] = lambda: _chain_case(1, 5, 7, 128, 512, 1, 423, residual=False)
CASES["chain128_multiround_b130"] = lambda: _chain_case(130, 28, 28, 128, 512, 1, 424)         # 797 tiles: two per workgroup
CASES["chain128_1x1_images"] = lambda: _chain_case(150, 1, 1, 128, 512, 1, 425)


def _chain_ds_case(B, Hh, Ww, N2, seed):
    """fused bottleneck tail whose shortcut is a 1x1 convolution of the 64-channel block input, multiplied inside the launch
    (first block of ResNet stage 1): against conv2 -> relu -> conv3 + shortcut convolution -> relu in fp32"""
    import hip_ops as H
    r = _rng(seed)
    C1 = 64
    y1 = _bf(r.standard_normal((B, Hh, Ww, C1)))                    # conv1 output = the tail's input
    x0 = _bf(r.standard_normal((B, Hh, Ww, 64)))                    # block input = the shortcut convolution's input
    k1 = (r.standard_normal((3, 3, C1, C1)) / math.sqrt(9 * C1)).astype(np.float32)
    s1, t1 = r.uniform(0.5, 1.5, C1).astype(np.float32), r.standard_normal(C1).astype(np.float32)
    k2 = (r.standard_normal((1, 1, C1, N2)) / math.sqrt(C1)).astype(np.float32)
    s2, t2 = r.uniform(0.5, 1.5, N2).astype(np.float32), r.standard_normal(N2).astype(np.float32)
    kd = (r.standard_normal((1, 1, 64, N2)) / 8).astype(np.float32)
    sd, td = r.uniform(0.5, 1.5, N2).astype(np.float32), r.standard_normal(N2).astype(np.float32)
    wt1, b1, K1, mode = pack.pack_conv(k1, s1, t1, C1)
    wt2, b2 = pack.pack_dense((k2.reshape(C1, N2) * s2.reshape(1, N2))[pack.chain_k_order(C1)], t2 + td)
    wds = pack.pack_chain_ds(kd.reshape(64, N2) * sd.reshape(1, N2))
    mid = O.conv2d(O.zero_pad2d(torch.from_numpy(y1), 1), torch.from_numpy(_bf(k1 * s1.reshape(1, 1, 1, -1))), None)
    mid = torch.from_numpy(_bf(torch.relu(mid + torch.from_numpy(t1)).numpy()))
    y = O.conv2d(mid, torch.from_numpy(_bf(k2 * s2.reshape(1, 1, 1, -1)))) + torch.from_numpy(t2)
    y = y + O.conv2d(torch.from_numpy(x0), torch.from_numpy(_bf(kd * sd.reshape(1, 1, 1, -1)))) + torch.from_numpy(td)
    y = torch.relu(y).numpy()
    got = H.conv_chain(H.dev_bf16(y1), H.dev_bits(wt1), H.dev_f32(b1), H.dev_bits(wt2), H.dev_f32(b2), None, KH=3, KW=3, stride=1,
                       pad=1, OH=Hh, OW=Ww, C1=C1, N2=N2, ds_x=H.dev_bf16(x0.reshape(-1, 64)), ds_w=H.dev_bits(wds))
    H.sync()
    return _err(_cpu(got).reshape(B, Hh, Ww, N2), y), TOL_BF16


CASES["chain_shortcut_conv_56x56_b3"] = lambda: _chain_ds_case(3, 56, 56, 256, 190)
CASES["chain_shortcut_conv_odd_31x17"] = lambda: _chain_ds_case(2, 31, 17, 256, 191)
CASES["chain_shortcut_conv_multiround_b40"] = lambda: _chain_ds_case(40, 56, 56, 256, 192)
CASES["chain_shortcut_conv_n512"] = lambda: _chain_ds_case(3, 14, 14, 512, 193)


# ---------------------------------------------------------------------------------------------
# fused transformer MLP (csrc/mlp.hip): LayerNorm -> fc1 -> act -> fc2 (-> LayerScale) -> + residual, hidden tensor in registers
# ---------------------------------------------------------------------------------------------
def _mlp_fused_case(M, seed, act="gelu", residual_is_x=True, layer_scale=False, x_offset=0.0):
    import hip_ops as H
    r = _rng(seed)
    Cc, Hd, eps = 128, 512, 1e-5
    x = _bf(r.standard_normal((M, Cc)) * r.uniform(0.2, 3.0, (M, 1)) + x_offset)
    gam, bet = r.uniform(0.5, 1.5, Cc).astype(np.float32), (0.3 * r.standard_normal(Cc)).astype(np.float32)
    k1 = (r.standard_normal((Cc, Hd)) / math.sqrt(Cc)).astype(np.float32)
    b1 = (0.2 * r.standard_normal(Hd)).astype(np.float32)
    k2 = (r.standard_normal((Hd, Cc)) / math.sqrt(Hd)).astype(np.float32)
    b2 = (0.2 * r.standard_normal(Cc)).astype(np.float32)
    ls = r.uniform(0.1, 1.0, Cc).astype(np.float32) if layer_scale else None
    res = x if residual_is_x else _bf(r.standard_normal((M, Cc)))
    w1, b1f, w2, b2f = pack.pack_mlp_fused(k1, b1, gam, bet, k2, b2, ls)
    # reference: the layer's own definition in fp64, on the bf16-rounded folded weights the launch multiplies with (the normalised
    # rows and the hidden activations are rounded to bf16 inside the launch: that is what the tolerance covers)
    xd = x.astype(np.float64)
    mean = xd.mean(axis=1, keepdims=True)
    var = ((xd - mean) ** 2).mean(axis=1, keepdims=True)
    w1r = pack.bf16_bits_to_f32(w1).astype(np.float64)                     # [Hd][C], gamma folded
    xhat = (xd - mean) / np.sqrt(var + eps)
    if _TIGHT:
        xhat = _bf(xhat.astype(np.float32)).astype(np.float64)            # the launch rounds the normalised rows to bf16 (the MFMA operand)
    hpre = xhat @ w1r.T + b1f.astype(np.float64)
    hact = _bf(O.activation(torch.from_numpy(hpre), act).numpy().astype(np.float32)).astype(np.float64)
    order = pack.chain_k_order(Hd)
    w2r = np.zeros((Cc, Hd))
    w2r[:, order] = pack.bf16_bits_to_f32(w2).astype(np.float64)           # undo the K permutation
    y = hact @ w2r.T + b2f.astype(np.float64) + res.astype(np.float64)
    got = H.mlp_fused(H.dev_bf16(x), H.dev_bits(w1), H.dev_f32(b1f), H.dev_bits(w2), H.dev_f32(b2f),
                      None if residual_is_x else H.dev_bf16(res), eps=eps, act=act)
    H.sync()
    return _err(_cpu(got), y), TOL_BF16


CASES["mlp_fused_one_tile"] = lambda: _mlp_fused_case(256, 300)
CASES["mlp_fused_ragged_77"] = lambda: _mlp_fused_case(77, 301)                                   # one partial tile
CASES["mlp_fused_ragged_3000"] = lambda: _mlp_fused_case(3000, 302, residual_is_x=False)          # 12 tiles, last one ragged
CASES["mlp_fused_multiround_70001"] = lambda: _mlp_fused_case(70001, 303)                         # 274 tiles: two rounds on some CUs
CASES["mlp_fused_layerscale_residual"] = lambda: _mlp_fused_case(3136 * 3, 304, residual_is_x=False, layer_scale=True)   # ConvNeXt block
CASES["mlp_fused_relu"] = lambda: _mlp_fused_case(1000, 305, act="relu")
CASES["mlp_fused_swish"] = lambda: _mlp_fused_case(1000, 306, act="swish")
CASES["mlp_fused_large_mean"] = lambda: _mlp_fused_case(2048, 307, x_offset=20.0)                 # mean >> spread: the c1 correction cancels
CASES["mlp_fused_swin_stage1_b8"] = lambda: _mlp_fused_case(8 * 3136, 308)


def _grouped_case(B, Hh, Ww, Cc, groups, stride, seed, act="relu"):
    import hip_ops as H
    r = _rng(seed)
    w = Cc // groups
    x = _bf(r.standard_normal((B, Hh, Ww, Cc)))
    k = (r.standard_normal((3, 3, w, Cc)) / math.sqrt(9 * w)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, Cc).astype(np.float32)
    shift = r.standard_normal(Cc).astype(np.float32)
    wfrag = pack.pack_grouped3x3(k, groups, scale)
    kf = _bf(k * scale.reshape(1, 1, 1, -1))
    y = O.conv2d(O.zero_pad2d(torch.from_numpy(x), 1), torch.from_numpy(kf), None, stride=stride, groups=groups)
    y = O.activation(y + torch.from_numpy(shift), act).numpy()
    got = H.grouped_conv3x3(H.dev_bf16(x), H.dev_bits(wfrag.reshape(-1, 8)), H.dev_f32(shift), stride, act)
    H.sync()
    return _err(_cpu(got), y), TOL_BF16


CASES["grouped3x3_c128_g32_56x56"] = lambda: _grouped_case(2, 56, 56, 128, 32, 1, 190)            # resnext50_32x4d layer 1: 4 channels per group
CASES["grouped3x3_c256_g32_s2_odd"] = lambda: _grouped_case(2, 29, 23, 256, 32, 2, 191)           # 8 per group, stride 2, odd sizes
CASES["grouped3x3_c1024_g32_7x7"] = lambda: _grouped_case(3, 7, 7, 1024, 32, 1, 192)              # 32 per group: dense 32 x 32 blocks
CASES["grouped3x3_c64_g4_tiny"] = lambda: _grouped_case(1, 3, 5, 64, 4, 1, 193, act="")           # 16 per group, partial pixel tile
CASES["grouped3x3_c96_g6_many_tiles"] = lambda: _grouped_case(9, 40, 40, 96, 6, 1, 194)           # 3 super-groups: last workgroup half empty


# ---------------------------------------------------------------------------------------------
# Tight cases: the fused product kernels at the layer shapes (and tiles) of the scored workloads, element-wise in bf16 ulps.
# Every reference above already rounds where the launch rounds (bf16 operands, the folded weights the launch multiplies with,
# one bf16 rounding at each tensor the unfused path would store); under _TIGHT the metric is _err_ulp and the bar is 1
# (= 2 ulps of the element + 2^-14 of the tensor's rms).
# ---------------------------------------------------------------------------------------------
def _tight(fn, stages=1):
    def run():
        global _TIGHT
        _TIGHT = stages
        try:
            out = fn()
        finally:
            _TIGHT = False
        errs = out if isinstance(out[0], tuple) else (out,)
        return max(e for e, _ in errs), 1.0
    return run


# ViT-B/16 (tile hint 21 = the 256 x 256 persistent tile these layers run on at batch 512)
CASES["tight_vit_b_qkv_ln_folded"] = _tight(lambda: _ln_gemm_case(2048, 768, 2304, "", 900, tile=21))
CASES["tight_vit_b_fc1_ln_folded_gelu"] = _tight(lambda: _ln_gemm_case(2048, 768, 3072, "gelu", 901, tile=21))
CASES["tight_vit_b_fc2_residual"] = _tight(lambda: _gemm_case(2048, 3072, 768, residual=True, tile=21, seed=902))
CASES["tight_vit_b_proj_residual"] = _tight(lambda: _gemm_case(2048, 768, 768, residual=True, tile=21, seed=903))
# ResNet-50 stage 1 and stem
CASES["tight_resnet50_chain_stage1"] = _tight(lambda: _chain_case(6, 56, 56, 64, 256, 1, 904), stages=2)
CASES["tight_resnet50_chain_stage1_shortcut_conv"] = _tight(lambda: _chain_ds_case(6, 56, 56, 256, 905), stages=2)
CASES["tight_resnet50_stem_conv_pool"] = _tight(lambda: _stem_pool_case(3, 224, 224, 906), stages=2)
CASES["tight_resnet50_chain_stage2"] = _tight(lambda: _chain_case(6, 28, 28, 128, 512, 1, 919), stages=2)
CASES["tight_resnet50_conv1_relu"] = _tight(lambda: _gemm_case(6 * 3136, 256, 64, act="relu", seed=907))
CASES["tight_resnet50_conv3_residual_relu"] = _tight(lambda: _gemm_case(6 * 784, 128, 512, act="relu", residual=True, act_after_res=True, seed=908))
CASES["tight_resnet50_conv3x3_stage2"] = _tight(lambda: _conv_case(6, 28, 28, 128, 128, 3, 1, 1, act="relu", seed=909))
# Swin-B stage 1 (and ConvNeXt-B stage 1): the one-launch MLP
CASES["tight_swin_b_mlp_fused_stage1"] = _tight(lambda: _mlp_fused_case(4 * 3136, 910), stages=2)
CASES["tight_convnext_b_mlp_fused_layerscale"] = _tight(lambda: _mlp_fused_case(2 * 3136, 911, residual_is_x=False, layer_scale=True), stages=2)
CASES["tight_swin_b_qkv_ln_folded_128"] = _tight(lambda: _ln_gemm_case(4 * 3136, 128, 384, "", 912))
# EfficientNet-B4 blocks 1-3: expansion + depthwise in one launch, swish epilogues, the SE-gated projection
CASES["tight_b4_expand_dw_24_144_k3s2"] = _tight(lambda: _expand_dw_case(1, 190, 190, 24, 144, 3, 2, "same", "swish", 913), stages=2)
CASES["tight_b4_expand_dw_32_192_k3s1"] = _tight(lambda: _expand_dw_case(1, 95, 95, 32, 192, 3, 1, "same", "swish", 914), stages=2)
CASES["tight_b4_expand_dw_32_192_k5s2"] = _tight(lambda: _expand_dw_case(1, 95, 95, 32, 192, 5, 2, "same", "swish", 915), stages=2)
CASES["tight_b4_expand_swish_56_336"] = _tight(lambda: _gemm_case(2 * 2304, 56, 336, act="swish", seed=916))
CASES["tight_b4_se_gated_projection_336_56"] = _tight(lambda: _se_scale_case(2, 2304, 336, 56, 917, bias=True, residual=True))
CASES["tight_b4_dwconv_k5_336"] = _tight(lambda: _dw_case(1, 48, 48, 336, 5, 1, "same", "swish", 918))


# ---------------------------------------------------------------------------------------------
# 3x3 / stride 1, 128 -> 128 channels on the input-strip kernel (csrc/conv_strip.hip, tile hint 31): against the oracle, and
# bit for bit against the implicit-GEMM tile it replaces (same operand roles, same order of the K reduction)
# ---------------------------------------------------------------------------------------------
def _strip_conv_case(B, Hh, Ww, seed, act="relu"):
    import hip_ops as H
    r = _rng(seed)
    C = 128
    x = _bf(r.standard_normal((B, Hh, Ww, C)))
    kern = (r.standard_normal((3, 3, C, C)) / math.sqrt(9 * C)).astype(np.float32)
    scale, shift = r.uniform(0.5, 1.5, C).astype(np.float32), r.standard_normal(C).astype(np.float32)
    wt, bias, K, mode = pack.pack_conv(kern, scale, shift, C)
    y = O.conv2d(O.zero_pad2d(torch.from_numpy(x), 1), torch.from_numpy(_bf(kern * scale.reshape(1, 1, 1, -1))), None) + torch.from_numpy(shift)
    y = O.activation(y, act).numpy()
    xd, wd, bd = H.dev_bf16(x), H.dev_bits(wt), H.dev_f32(bias)
    conv = dict(mode=mode, B=B, H=Hh, W=Ww, Cin=C, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, OH=Hh, OW=Ww)
    got = H.gemm(xd, wd, C, K, bias=bd, act=act, conv=conv, tile_hint=31)
    other = H.gemm(xd, wd, C, K, bias=bd, act=act, conv=conv, tile_hint=24)
    H.sync()
    if not torch.equal(got.view(torch.int16), other.view(torch.int16)):
        return float("inf"), TOL_BF16
    return _err(_cpu(got).reshape(B, Hh, Ww, C), y), TOL_BF16


CASES["strip_conv_28x28_b6"] = lambda: _strip_conv_case(6, 28, 28, 400)                     # ResNet-50 stage 2: 37 tiles, the last one ragged
CASES["strip_conv_odd_17x23"] = lambda: _strip_conv_case(3, 17, 23, 401)                    # every border case of the tap mask, images inside tiles
CASES["strip_conv_31_wide"] = lambda: _strip_conv_case(2, 9, 31, 402)                       # the widest row the strip holds
CASES["strip_conv_tiny_5x7"] = lambda: _strip_conv_case(1, 5, 7, 403)                       # one partial tile, strip mostly out of range
CASES["strip_conv_1x1_images"] = lambda: _strip_conv_case(200, 1, 1, 404)                   # every tap but the centre masked
CASES["strip_conv_multiround_b130"] = lambda: _strip_conv_case(130, 28, 28, 405)            # 797 tiles: two per workgroup on most CUs
CASES["strip_conv_swish_14x14"] = lambda: _strip_conv_case(9, 14, 14, 406, act="swish")
CASES["tight_resnet50_conv3x3_stage2_strip"] = _tight(lambda: _strip_conv_case(6, 28, 28, 407))


def run_case(name):
    out = CASES[name]()
    err, tol = out
    return float(err), float(tol)
