"""ctypes binding of libtfimm_hip.so (C ABI: include/tfimm_hip.h).

The library is the ONLY compute path of this package: if it cannot be loaded the import of
this module raises -- there is deliberately no CPU or PyTorch fallback.
"""
import ctypes as C
import os

# torch ships its own libamdhip64/libhsa-runtime64.  It must be loaded FIRST so that
# libtfimm_hip.so's DT_NEEDED libamdhip64.so.7 resolves to that already-loaded runtime: two HIP
# runtimes in one process do not share devices/streams (launches fail with hipErrorNoDevice).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# TFIMM_HIP_LIB: another build of the same library (kernel A/B comparisons); there is no non-HIP fallback
LIB_PATH = os.environ.get("TFIMM_HIP_LIB") or os.path.join(_HERE, "libtfimm_hip.so")

ACT = {
    "": 0, "linear": 0, "none": 0, None: 0,
    "relu": 1, "gelu": 2, "swish": 3, "sigmoid": 4, "relu6": 5, "tanh": 6,
}
A_DENSE, A_CONV, A_CONV_C4 = 0, 1, 2


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("wt", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("out", C.c_void_p), ("a_scale", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldr", C.c_int32), ("ldc", C.c_int32),
        ("out_f32", C.c_int32), ("act", C.c_int32), ("act_after_res", C.c_int32),
        ("res_mod", C.c_int32),
        ("remap_in", C.c_int32), ("remap_out", C.c_int32), ("remap_off", C.c_int32),
        ("mode", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
        ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("rows_per_image", C.c_int32), ("tile_hint", C.c_int32), ("stride_w", C.c_int32),
        ("pix_pitch", C.c_int32),
        ("ln_stats", C.c_void_p), ("ln_c1", C.c_void_p),
        # ABI v4: a second A operand (the shortcut convolution of a residual block folded into its last 1x1 convolution)
        ("a2", C.c_void_p), ("K2", C.c_int32), ("lda2", C.c_int32),
        ("a2_stride", C.c_int32), ("a2_H", C.c_int32), ("a2_W", C.c_int32), ("a2_OH", C.c_int32), ("a2_OW", C.c_int32),
        ("a2_window", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("out", C.c_void_p), ("rel_bias", C.c_void_p),
        ("batch", C.c_int32), ("n_tokens", C.c_int32), ("heads", C.c_int32), ("hd", C.c_int32),
        ("scale", C.c_float),
        ("window", C.c_int32), ("shift", C.c_int32), ("res_h", C.c_int32), ("res_w", C.c_int32),
        ("bias_log2", C.c_void_p),
    ]


class ThaDesc(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("out", C.c_void_p),
        ("proj_l_w", C.c_void_p), ("proj_l_b", C.c_void_p), ("proj_w_w", C.c_void_p), ("proj_w_b", C.c_void_p),
        ("batch", C.c_int32), ("n_tokens", C.c_int32), ("heads", C.c_int32), ("hd", C.c_int32),
        ("scale", C.c_float), ("proj_dev", C.c_void_p),
    ]


class StemDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wt", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("batch", C.c_int32), ("Hp", C.c_int32), ("Wp2", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("ldw", C.c_int32),
        ("in_dtype", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
    ]


class ChainDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("residual", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
        ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("C1", C.c_int32), ("N2", C.c_int32),
        ("ldw1", C.c_int32), ("ldw2", C.c_int32), ("ldr", C.c_int32), ("ldc", C.c_int32),
        ("act1", C.c_int32), ("act2", C.c_int32),
        ("ds_x", C.c_void_p), ("ds_w", C.c_void_p), ("ds_cin", C.c_int32),
    ]


class ExpandDwDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("wdw", C.c_void_p), ("b2", C.c_void_p),
        ("y", C.c_void_p), ("sum_out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("C", C.c_int32), ("Cpad", C.c_int32),
        ("k", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("OH", C.c_int32),
        ("OW", C.c_int32), ("act1", C.c_int32), ("act2", C.c_int32),
        ("stem", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32),
    ]


class MlpDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
                ("b2", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int64),
                ("C", C.c_int32), ("hidden", C.c_int32), ("act", C.c_int32), ("eps", C.c_float)]


# name -> (restype, argtypes); every symbol declared in include/tfimm_hip.h
_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "tfimm_hip_abi_version": (_i, []),
    "tfimm_hip_last_error": (C.c_char_p, []),
    "tfimm_hip_device_info": (_i, [_i, C.c_char_p, _i]),
    "tfimm_hip_gemm": (_i, [C.POINTER(GemmDesc), _vp]),
    "tfimm_hip_conv_chain": (_i, [C.POINTER(ChainDesc), _vp]),
    "tfimm_hip_mlp_fused": (_i, [C.POINTER(MlpDesc), _vp]),
    "tfimm_hip_expand_dwconv": (_i, [C.POINTER(ExpandDwDesc), _vp]),
    "tfimm_hip_stem_conv_pool": (_i, [C.POINTER(StemDesc), _vp]),
    "tfimm_hip_cast_input": (_i, [_vp, _i, _vp, _i64, _i, _i, _vp]),
    "tfimm_hip_cast_input_pad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_preprocess_input": (_i, [_vp, _vp, _i64, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "tfimm_hip_preprocess_input_pad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), _vp]),
    "tfimm_hip_row_stats": (_i, [_vp, _vp, _i64, _i, _i64, _f, _vp]),
    "tfimm_hip_layernorm": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i64, _i64, _f, _vp]),
    "tfimm_hip_attention": (_i, [C.POINTER(AttnDesc), _vp]),
    "tfimm_hip_talking_heads_attention": (_i, [C.POINTER(ThaDesc), _vp]),
    "tfimm_hip_class_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_copy_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_maxpool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_mean_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "tfimm_hip_bcast_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "tfimm_hip_dwconv": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_se_gate": (_i, [_vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_scale_channels": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tfimm_hip_patch_merge_ln": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "tfimm_hip_attention_probs": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "tfimm_hip_group_norm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "tfimm_hip_blur_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_avg_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_eca_gate": (_i, [_vp, _f, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tfimm_hip_grouped_conv3x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_bias_act": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    # program-level entry points (csrc/plan.hip): a serialised plan (graph.Plan.export) run without Python host logic
    "tfimm_hip_plan_query": (_i, [_vp, C.c_size_t, _vp]),
    "tfimm_hip_plan_create": (_i, [_vp, C.c_size_t, _vp, _vp, C.POINTER(_vp)]),
    "tfimm_hip_plan_forward": (_i, [_vp, _vp, _i, _vp]),
    "tfimm_hip_plan_output": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i)]),
    "tfimm_hip_plan_destroy": (_i, [_vp]),
    # float32 verification path (csrc/ref32.hip): the bf16 signatures with float tensors
    "tfimm_hip_ref_gemm": (_i, [C.POINTER(GemmDesc), _vp]),
    "tfimm_hip_ref_cast_input": (_i, [_vp, _i, _vp, _i64, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp]),
    "tfimm_hip_ref_layernorm": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i64, _i64, _f, _vp]),
    "tfimm_hip_ref_patch_merge_ln": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "tfimm_hip_ref_copy_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_ref_bcast_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "tfimm_hip_ref_mean_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "tfimm_hip_ref_scale_channels": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tfimm_hip_ref_maxpool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_ref_avg_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_ref_blur_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_ref_dwconv": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_ref_group_norm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "tfimm_hip_ref_attention": (_i, [C.POINTER(AttnDesc), _vp]),
    "tfimm_hip_ref_attention_probs": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "tfimm_hip_ref_talking_heads_attention": (_i, [C.POINTER(ThaDesc), _vp]),
    "tfimm_hip_ref_class_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tfimm_hip_memset_async": (_i, [_vp, _i, C.c_size_t, _vp]),
}


class PlanInfo(C.Structure):
    _fields_ = [("workspace_bytes", C.c_uint64), ("batch", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("in_c", C.c_int32), ("n_calls", C.c_int32), ("n_outputs", C.c_int32)]


class HipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C tensorflow-image-models_amd/csrc -j8` "
            "(or __graft_entry__.build()). tfimm has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    # 4 = this tree.  3 (tfimm_gemm_desc without the second A operand) is accepted ONLY for a library named by TFIMM_HIP_LIB --
    # an older build in a kernel A/B; the lowering then does not emit what that build cannot run (ABI below)
    v = lib.tfimm_hip_abi_version()
    if v != 4 and not (v == 3 and os.environ.get("TFIMM_HIP_LIB")):
        raise ImportError("libtfimm_hip.so ABI version mismatch")
    return lib


lib = _load()
ABI = lib.tfimm_hip_abi_version()


_dp_lib = None


def dp_lib():
    """libtfimm_hip_dp.so (include/tfimm_hip_dp.h: the data-parallel exchange behind a C ABI), loaded on first use -- it
    links RCCL, which a single-GPU forward never needs."""
    global _dp_lib
    if _dp_lib is None:
        path = os.environ.get("TFIMM_HIP_DP_LIB") or os.path.join(_HERE, "libtfimm_hip_dp.so")
        if not os.path.exists(path):
            raise HipError(f"{path} not found: build it with `make -C tensorflow-image-models_amd/csrc`")
        L = C.CDLL(path)
        L.tfimm_hip_dp_abi_version.restype = C.c_int
        L.tfimm_hip_dp_last_error.restype = C.c_char_p
        L.tfimm_hip_dp_shard_bounds.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.tfimm_hip_dp_unique_id.argtypes = [C.c_void_p, C.c_size_t]
        L.tfimm_hip_dp_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.tfimm_hip_dp_world.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.tfimm_hip_dp_all_gather_logits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.tfimm_hip_dp_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.tfimm_hip_dp_destroy.argtypes = [C.c_void_p]
        for f in ("shard_bounds", "unique_id", "create", "world", "all_gather_logits", "forward", "destroy"):
            getattr(L, "tfimm_hip_dp_" + f).restype = C.c_int
        assert L.tfimm_hip_dp_abi_version() == 1
        _dp_lib = L
    return _dp_lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib.tfimm_hip_last_error().decode("utf-8", "replace")
        raise HipError(f"{what or 'tfimm_hip call'} failed (rc={rc}): {msg}")


# ---- roctx markers (TFIMM_ROCTX=1, graph.Plan.run): libroctx64 ships with ROCm; absent library = no markers, never an error
_roctx = None


def _roctx_lib():
    global _roctx
    if _roctx is None:
        _roctx = False
        # rocprofv3 (rocprofiler-sdk) traces its own roctx library; libroctx64 is the roctracer-era one
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                cand = C.CDLL(name)
                cand.roctxRangePushA.argtypes = [C.c_char_p]
                cand.roctxRangePushA.restype = C.c_int
                cand.roctxRangePop.restype = C.c_int
                _roctx = cand
                break
            except (OSError, AttributeError):
                continue
    return _roctx


def roctx_push(label: str) -> None:
    lib_ = _roctx_lib()
    if lib_:
        lib_.roctxRangePushA(label.encode("utf-8", "replace"))


def roctx_pop() -> None:
    lib_ = _roctx_lib()
    if lib_:
        lib_.roctxRangePop()
