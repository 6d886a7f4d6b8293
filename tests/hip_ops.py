"""Thin torch-tensor wrappers over the C ABI for op-level parity tests (test helper).

Every function takes/returns CUDA torch tensors (bf16 activations) and calls
libtfimm_hip.so through tfimm.engine.ffi -- the same entry points the engine uses.
"""
import ctypes as C

import numpy as np
import torch

from tfimm.engine import ffi, pack

lib = ffi.lib
# On a box without a GPU the wrappers still marshal every argument and call into the library
# (the launch then fails with a HIP error): lets the CPU suite dry-run the call plumbing.
DEV = "cuda" if torch.cuda.is_available() else "cpu"


def stream():
    if DEV == "cpu":
        return C.c_void_p(0)
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def sync():
    if DEV == "cuda":
        torch.cuda.synchronize()


def dev_bf16(a) -> torch.Tensor:
    t = torch.as_tensor(np.asarray(a), dtype=torch.float32)
    return t.to(torch.bfloat16).to(DEV).contiguous()


def dev_f32(a) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).to(DEV).contiguous()


def dev_bits(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.view(np.int16).copy()).to(DEV).view(torch.bfloat16)


def ptr(t):
    return None if t is None else t.data_ptr()


def gemm(a, wt, N, K, *, M=None, bias=None, residual=None, out=None, act="", act_after_res=False,
         out_f32=False, lda=None, ldc=None, ldr=None, res_mod=0, remap=None, conv=None, a_scale=None,
         rows_per_image=0, tile_hint=0, out_rows=None, a_byte_offset=0, out_byte_offset=0, ln_stats=None, ln_c1=None,
         a2=None, a2_geom=None):
    """conv: dict(mode, B, H, W, Cin, KH, KW, stride, pad_t, pad_l, OH, OW).
    a2: second A operand [rows2][K2] (tfimm_gemm_desc::a2); a2_geom = (stride, H, W, OH, OW[, window]) for a strided row view (window > 1: w x w taps)."""
    d = ffi.GemmDesc()
    if conv is None:
        M = M if M is not None else a.shape[0]
        d.mode = 0
        d.lda = lda if lda is not None else a.shape[-1]
    else:
        d.mode = conv["mode"]
        d.B, d.H, d.W, d.Cin = conv["B"], conv["H"], conv["W"], conv["Cin"]
        d.KH, d.KW, d.stride = conv["KH"], conv["KW"], conv["stride"]
        d.pad_t, d.pad_l, d.OH, d.OW = conv["pad_t"], conv["pad_l"], conv["OH"], conv["OW"]
        d.stride_w = conv.get("stride_w", 0)
        d.pix_pitch = conv.get("pix_pitch", 0)
        M = conv["B"] * conv["OH"] * conv["OW"]
    d.M, d.N, d.K = M, N, K
    d.ldw = wt.shape[1]
    rows = out_rows if out_rows is not None else M
    if out is None:
        out = torch.empty(rows, ldc or N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=DEV)
        if ldc or remap:
            out.zero_()
    d.a, d.wt, d.bias, d.residual, d.out = ptr(a), ptr(wt), ptr(bias), ptr(residual), ptr(out)
    if a_byte_offset:
        d.a = ptr(a) + a_byte_offset
    if out_byte_offset:
        d.out = ptr(out) + out_byte_offset
    d.ldc = ldc or N
    d.ldr = ldr or (residual.shape[-1] if residual is not None else 0)
    d.out_f32 = 1 if out_f32 else 0
    d.act = ffi.ACT[act]
    d.act_after_res = 1 if act_after_res else 0
    d.res_mod = res_mod
    if remap:
        d.remap_in, d.remap_out, d.remap_off = remap
    if a_scale is not None:
        d.a_scale = ptr(a_scale)
        d.rows_per_image = rows_per_image
    if ln_stats is not None:
        d.ln_stats, d.ln_c1 = ptr(ln_stats), ptr(ln_c1)
    if a2 is not None:
        d.a2, d.K2, d.lda2 = ptr(a2), a2.shape[-1], a2.shape[-1]
        d.a2_stride = 1
        if a2_geom is not None:
            d.a2_stride, d.a2_H, d.a2_W, d.a2_OH, d.a2_OW = a2_geom[:5]
            d.a2_window = a2_geom[5] if len(a2_geom) > 5 else 0
    if tile_hint == "table":        # what the engine would launch for this shape (tfimm/engine/gemm_tune.json)
        from tfimm.engine import tune
        tile_hint = tune.lookup(d)
    d.tile_hint = tile_hint
    ffi.check(lib.tfimm_hip_gemm(C.byref(d), stream()), "gemm")
    return out


def row_stats(x, eps, rows=None, d=None, xs=None):
    rows = rows if rows is not None else x.shape[0]
    d = d if d is not None else x.shape[-1]
    st = torch.empty(rows, 2, dtype=torch.float32, device=DEV)
    ffi.check(lib.tfimm_hip_row_stats(ptr(x), ptr(st), rows, d, xs or d, float(eps), stream()), "row_stats")
    return st


def layernorm(x, gamma, beta, eps, rows=None, d=None, xs=None, ys=None, out=None):
    rows = rows if rows is not None else x.shape[0]
    d = d if d is not None else x.shape[-1]
    out = out if out is not None else torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_layernorm(ptr(x), ptr(out), ptr(gamma), ptr(beta), rows, d, xs or d, ys or d,
                                      float(eps), stream()), "layernorm")
    return out


def attention(qkv, batch, n_tokens, heads, hd, scale, window=0, shift=0, res=(0, 0), rel_bias=None, bias_log2=None):
    out = torch.empty(batch * n_tokens, heads * hd, dtype=torch.bfloat16, device=DEV)
    d = ffi.AttnDesc()
    d.qkv, d.out, d.rel_bias = ptr(qkv), ptr(out), ptr(rel_bias)
    d.bias_log2 = ptr(bias_log2)
    d.batch, d.n_tokens, d.heads, d.hd, d.scale = batch, n_tokens, heads, hd, float(scale)
    d.window, d.shift, d.res_h, d.res_w = window, shift, res[0], res[1]
    ffi.check(lib.tfimm_hip_attention(C.byref(d), stream()), "attention")
    return out


def talking_heads_attention(qkv, batch, n_tokens, heads, hd, scale, wl, bl, ww, bw, use_dev=True):
    """wl, bl, ww, bw: HOST numpy fp32 arrays (the C ABI takes the two head-mixing layers by host pointer); ``use_dev``: also
    hand over their packed DEVICE copy (tfimm_tha_desc.proj_dev: what plans do)."""
    out = torch.empty(batch * n_tokens, heads * hd, dtype=torch.bfloat16, device=DEV)
    d = ffi.ThaDesc()
    d.qkv, d.out = ptr(qkv), ptr(out)
    host = [np.ascontiguousarray(a, dtype=np.float32) for a in (wl, bl, ww, bw)]
    dev = torch.from_numpy(np.concatenate([a.reshape(-1) for a in host])).to(DEV) if use_dev else None
    d.proj_dev = ptr(dev)
    d.proj_l_w, d.proj_l_b, d.proj_w_w, d.proj_w_b = (a.ctypes.data for a in host)
    d.batch, d.n_tokens, d.heads, d.hd, d.scale = batch, n_tokens, heads, hd, float(scale)
    ffi.check(lib.tfimm_hip_talking_heads_attention(C.byref(d), stream()), "talking_heads_attention")
    return out


def class_attention(q, kv, batch, n_tokens, heads, hd):
    out = torch.empty(batch, heads * hd, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_class_attention(ptr(q), ptr(kv), ptr(out), batch, n_tokens, heads, hd, q.shape[-1],
                                            kv.shape[-1], heads * hd, stream()), "class_attention")
    return out


def copy_rows(src, dst, dst_row0):
    B, R, D = src.shape
    ffi.check(lib.tfimm_hip_copy_rows(ptr(src), ptr(dst), B, R, dst.shape[1], dst_row0, D, stream()), "copy_rows")
    return dst


def cast_input(x, c_out):
    B, H, W, Cin = x.shape
    out = torch.empty(B, H, W, c_out, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_cast_input(ptr(x), 1 if x.dtype == torch.bfloat16 else 0, ptr(out), B * H * W, Cin,
                                       c_out, stream()), "cast_input")
    return out


def cast_input_pad(x, pad):
    """pad = (top, bottom, left, right); returns the zero-bordered 4-channel bf16 image."""
    B, H, W, Cin = x.shape
    pt, pb, pl, pr = pad
    out = torch.empty(B, H + pt + pb, W + pl + pr, 4, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_cast_input_pad(ptr(x), 1 if x.dtype == torch.bfloat16 else 0, ptr(out), B, H, W, Cin,
                                           pt, pb, pl, pr, stream()), "cast_input_pad")
    return out


def _norm_arrays(mean, std):
    import ctypes as C
    return (C.c_float * len(mean))(*[float(v) for v in mean]), (C.c_float * len(std))(*[float(v) for v in std])


def preprocess_input(u8, c_out, mean, std):
    B, H, W, Cin = u8.shape
    out = torch.empty(B, H, W, c_out, dtype=torch.bfloat16, device=DEV)
    m, s = _norm_arrays(mean, std)
    ffi.check(lib.tfimm_hip_preprocess_input(ptr(u8), ptr(out), B * H * W, Cin, c_out, m, s, stream()), "preprocess_input")
    return out


def preprocess_input_pad(u8, pad, mean, std):
    B, H, W, Cin = u8.shape
    pt, pb, pl, pr = pad
    out = torch.empty(B, H + pt + pb, W + pl + pr, 4, dtype=torch.bfloat16, device=DEV)
    m, s = _norm_arrays(mean, std)
    ffi.check(lib.tfimm_hip_preprocess_input_pad(ptr(u8), ptr(out), B, H, W, Cin, pt, pb, pl, pr, m, s, stream()),
              "preprocess_input_pad")
    return out


def stem_conv_pool(x, wt, bias, B, Hp, Wp2, OH, OW, raw=None):
    """x: the zero-bordered 4-channel bf16 image (cast_input_pad), or with raw=(H, W, pad_t, pad_l) the caller's
    (B, H, W, 3) image in bf16 / float32; returns (B, PH, PW, 64) bf16."""
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    out = torch.empty(B, PH, PW, 64, dtype=torch.bfloat16, device=DEV)
    d = ffi.StemDesc()
    d.x, d.wt, d.bias, d.out = ptr(x), ptr(wt), ptr(bias), ptr(out)
    d.batch, d.Hp, d.Wp2, d.OH, d.OW, d.ldw = B, Hp, Wp2, OH, OW, wt.shape[1]
    if raw is not None:
        d.in_dtype = 1 if x.dtype == torch.bfloat16 else 2
        d.H, d.W, d.pad_t, d.pad_l = raw
    ffi.check(lib.tfimm_hip_stem_conv_pool(d, stream()), "stem_conv_pool")
    return out


def maxpool(x, k, stride, pad):
    B, H, W, Cc = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty(B, OH, OW, Cc, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_maxpool(ptr(x), ptr(out), B, H, W, Cc, k, stride, pad, OH, OW, stream()), "maxpool")
    return out


def mean_rows(x, out_f32=False):
    B, R, Cc = x.shape
    out = torch.empty(B, Cc, dtype=torch.float32 if out_f32 else torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_mean_rows(ptr(x), ptr(out), B, R, Cc, 1 if out_f32 else 0, stream()), "mean_rows")
    return out


def bcast_rows(src, dst, B, n_rows, d, dst_rows):
    ffi.check(lib.tfimm_hip_bcast_rows(ptr(src), ptr(dst), B, n_rows, d, dst_rows, stream()), "bcast_rows")
    return dst


def dwconv(x, w, bias, k, stride, pad_t, pad_l, OH, OW, act="", want_sums=False):
    B, H, W, Cc = x.shape
    out = torch.empty(B, OH, OW, Cc, dtype=torch.bfloat16, device=DEV)
    sums = torch.zeros(B, Cc, dtype=torch.int64, device=DEV) if want_sums else None      # fixed point, 2^-20 units
    ffi.check(lib.tfimm_hip_dwconv(ptr(x), ptr(w), ptr(bias), ptr(out), ptr(sums), B, H, W, Cc, k, stride, pad_t,
                                   pad_l, OH, OW, ffi.ACT[act], stream()), "dwconv")
    return out, sums


def expand_dwconv(x, w1frag, b1, wdw, b2, Cexp, k, stride, pad_t, pad_l, OH, OW, act="", want_sums=False, stem_hw=None):
    B, H, W, Cin = x.shape
    out = torch.empty(B, OH, OW, Cexp, dtype=torch.bfloat16, device=DEV)
    sums = torch.zeros(B, Cexp, dtype=torch.int64, device=DEV) if want_sums else None    # fixed point, 2^-20 units
    d = ffi.ExpandDwDesc()
    d.x, d.w1, d.b1, d.wdw, d.b2, d.y, d.sum_out = ptr(x), ptr(w1frag), ptr(b1), ptr(wdw), ptr(b2), ptr(out), ptr(sums)
    d.B, d.H, d.W, d.Cin, d.C, d.Cpad = B, H, W, Cin, Cexp, b1.numel()
    d.k, d.stride, d.pad_t, d.pad_l, d.OH, d.OW = k, stride, pad_t, pad_l, OH, OW
    d.act1 = d.act2 = ffi.ACT[act]
    if stem_hw is not None:        # x is the zero-bordered 4-channel image; stem_hw = the stem convolution's output size
        d.stem, d.img_h, d.img_w = 1, H, W
        d.H, d.W = stem_hw
    ffi.check(lib.tfimm_hip_expand_dwconv(C.byref(d), stream()), "expand_dwconv")
    return out, sums


def sums_to_float(sums):
    """int64 fixed-point squeeze sums (2^-20 units) -> float64 numpy"""
    return sums.cpu().numpy().astype(np.float64) / 2.0 ** 20


def se_gate(sums, inv_count, w1, b1, w2, b2, act, gate_act="sigmoid"):
    B, Cc = sums.shape
    rd = w1.shape[0]
    gate = torch.empty(B, Cc, dtype=torch.float32, device=DEV)
    ffi.check(lib.tfimm_hip_se_gate(ptr(sums), 1 if sums.dtype == torch.int64 else 0, float(inv_count), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(gate), B, Cc,
                                    rd, ffi.ACT[act], ffi.ACT[gate_act], stream()), "se_gate")
    return gate


def scale_channels(x, gate, residual=None, relu_after=False):
    B, R, Cc = x.shape
    out = torch.empty_like(x)
    ffi.check(lib.tfimm_hip_scale_channels(ptr(x), ptr(gate), ptr(residual), ptr(out), B, R, Cc,
                                           1 if relu_after else 0, stream()), "scale_channels")
    return out


def patch_merge_ln(x, gamma, beta, H, W, eps):
    B, L, Cc = x.shape
    out = torch.empty(B, (H // 2) * (W // 2), 4 * Cc, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_patch_merge_ln(ptr(x), ptr(out), ptr(gamma), ptr(beta), B, H, W, Cc, float(eps),
                                           stream()), "patch_merge_ln")
    return out


def attention_probs(qkv, B, n, heads, hd, scale):
    out = torch.empty(B, heads, n, n, dtype=torch.float32, device=DEV)
    ffi.check(lib.tfimm_hip_attention_probs(ptr(qkv), ptr(out), B, n, heads, hd, float(scale), stream()), "attention_probs")
    return out


def group_norm(x, gamma, beta, groups, eps, act="", residual=None, act_after=""):
    B, R, Cc = x.shape
    out = torch.empty_like(x)
    ws = torch.full((B, groups, 2), 7, dtype=torch.int64, device=DEV)         # poisoned: the entry point zeroes it
    ffi.check(lib.tfimm_hip_group_norm(ptr(x), ptr(gamma), ptr(beta), ptr(residual), ptr(out), ptr(ws), B, R, Cc, groups,
                                       float(eps), ffi.ACT[act], ffi.ACT[act_after], stream()), "group_norm")
    return out


def blur_pool(x, stride):
    B, H, W, Cc = x.shape
    p = (3 + stride) // 2 - 1
    OH, OW = (H + 2 * p - 3) // stride + 1, (W + 2 * p - 3) // stride + 1
    out = torch.empty(B, OH, OW, Cc, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_blur_pool(ptr(x), ptr(out), B, H, W, Cc, stride, stream()), "blur_pool")
    return out


def avg_pool(x, k, stride):
    B, H, W, Cc = x.shape
    out = torch.empty(B, -(-H // stride), -(-W // stride), Cc, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_avg_pool(ptr(x), ptr(out), B, H, W, Cc, k, stride, stream()), "avg_pool")
    return out


def eca_gate(sums, inv_count, w, gate_act="sigmoid"):
    B, Cc = sums.shape
    gate = torch.empty(B, Cc, dtype=torch.float32, device=DEV)
    ffi.check(lib.tfimm_hip_eca_gate(ptr(sums), float(inv_count), ptr(w), ptr(gate), B, Cc, int(w.shape[0]),
                                     ffi.ACT[gate_act], stream()), "eca_gate")
    return gate


def conv_chain(x, wt1, b1, wt2, b2, residual, *, KH, KW, stride, pad, OH, OW, C1, N2, act1="relu", act2="relu", ds_x=None,
               ds_w=None):
    B, H, W, Cin = x.shape
    d = ffi.ChainDesc()
    out = torch.empty(B * OH * OW, N2, dtype=torch.bfloat16, device=DEV)
    d.x, d.w1, d.b1, d.w2, d.b2, d.residual, d.out = ptr(x), ptr(wt1), ptr(b1), ptr(wt2), ptr(b2), ptr(residual), ptr(out)
    d.B, d.H, d.W, d.Cin, d.KH, d.KW, d.stride, d.pad_t, d.pad_l, d.OH, d.OW = B, H, W, Cin, KH, KW, stride, pad, pad, OH, OW
    d.C1, d.N2, d.ldw1, d.ldw2, d.ldr, d.ldc = C1, N2, wt1.shape[1], wt2.shape[1], N2, N2
    d.act1, d.act2 = ffi.ACT[act1], ffi.ACT[act2]
    if ds_x is not None:
        d.ds_x, d.ds_w, d.ds_cin = ptr(ds_x), ptr(ds_w), ds_x.shape[-1]
    ffi.check(lib.tfimm_hip_conv_chain(C.byref(d), stream()), "conv_chain")
    return out


def mlp_fused(x, w1, b1, w2, b2, residual=None, *, eps=1e-5, act="gelu"):
    M, Cc = x.shape
    d = ffi.MlpDesc()
    out = torch.empty(M, Cc, dtype=torch.bfloat16, device=DEV)
    d.x, d.w1, d.b1, d.w2, d.b2, d.out = ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(out)
    d.residual = ptr(x if residual is None else residual)
    d.M, d.C, d.hidden, d.act, d.eps = M, Cc, w1.shape[0], ffi.ACT[act], eps
    ffi.check(lib.tfimm_hip_mlp_fused(C.byref(d), stream()), "mlp_fused")
    return out


def grouped_conv3x3(x, wfrag, bias, stride, act=""):
    B, H, W, Cc = x.shape
    OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    out = torch.empty(B, OH, OW, Cc, dtype=torch.bfloat16, device=DEV)
    ffi.check(lib.tfimm_hip_grouped_conv3x3(ptr(x), ptr(wfrag), ptr(bias), ptr(out), B, H, W, Cc, stride, ffi.ACT[act],
                                            stream()), "grouped_conv3x3")
    return out
