"""GEMM tile selection table.

``tfimm_hip_gemm`` picks a tile shape with a static cost model unless the descriptor carries a
``tile_hint``.  This module keeps a table ``problem shape -> tile_hint`` that is
  * loaded from ``gemm_tune.json`` next to this file (measured on an MI355X by
    ``tools/tune_gemm.py``; committed so that runs are reproducible), and
  * extended at plan-build time when autotuning is on (``TFIMM_AUTOTUNE=1`` or
    ``tune.enable_autotune()``): ``Plan.autotune`` times every candidate on the plan's own
    buffers before the first forward.
Hints: 0 = library cost model, 11..16 = one-tile-per-workgroup LDS-DMA tiles, 21..26 =
persistent LDS-DMA tiles (ids: 256x256, 256x128, 128x128, 256x64, 128x64, 128x256; 27 = 256x64 with
64x64 wave tiles; 28 = 256x256 with the four-stage ring of gemm_pipe_kernel.h; 29 = 256x32 for narrow outputs;
30 = 256x128 with two co-resident four-wave workgroups per CU, gemm_duo_kernel.h; 31 = the input-strip kernel for
3x3 / stride 1 convolutions of 128 -> 128 channels, csrc/conv_strip.hip -- any other shape falls back to the cost model).
"""
import json
import os

# TFIMM_GEMM_TUNE: another table file (A/B of two library builds, each with the table tuned for it: tools/ab_lib.sh)
_PATH = os.environ.get("TFIMM_GEMM_TUNE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tune.json")
CANDIDATES = (0, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 11, 12, 13, 14, 15, 16)
# a layer with a folded LayerNormalization runs on the persistent tiles only (the library ignores any other hint for it)
LN_CANDIDATES = (0, 21, 22, 23, 24, 25, 26, 27, 29, 30)
TABLE = {}
_autotune = os.environ.get("TFIMM_AUTOTUNE", "0") == "1"


def enable_autotune(on: bool = True):
    global _autotune
    _autotune = bool(on)


def autotune_enabled() -> bool:
    return _autotune


def key_of(d) -> str:
    """Everything that changes the kernel's work: GEMM extents, operand strides, conv geometry and
    the epilogue flavour (residual / fp32 output change the epilogue's memory traffic) and the SE-gate
    prologue.  A layer with a folded LayerNormalization (``ln_stats``: other kernel flavour, longer epilogue, only
    the persistent tiles) gets an entry of its own, ``...:ln``."""
    key = ":".join(str(int(v)) for v in (
        d.mode, d.M, d.N, d.K, d.lda, d.ldc, d.H, d.W, d.Cin, d.KH, d.KW, d.stride,
        1 if d.residual else 0, d.out_f32, d.act, 1 if d.a_scale else 0))
    if getattr(d, "a2", None):          # a second A operand (ABI v4): K2 more channels, read at a2_stride
        key += f":d{int(d.K2)}s{int(d.a2_stride)}" + (f"w{int(d.a2_window)}" if getattr(d, "a2_window", 0) > 1 else "")
    return key + ":ln" if getattr(d, "ln_stats", None) else key


# a layer with an SE gate on its A operand: hints 11..16 send it to the register-staged family (the LDS-DMA family has no
# gate flavour), which scales every A element once while it stages it -- for the narrow project layers of EfficientNet that
# beats the persistent kernels (every wave there scales the fragments it reads); its six tiles are candidates of their own
SCALE_CANDIDATES = CANDIDATES + (1, 2, 3, 4, 5, 6)


def strip_shape(d) -> bool:
    """The one shape the input-strip kernel (hint 31, csrc/conv_strip.hip) is built for -- what ``tfimm_hip_gemm`` checks
    before it honours the hint: 3 x 3 / stride 1 / pad 1 convolution of 128 -> 128 channels, rows of at most 31 pixels, same
    output size, no residual, bf16 output.  For any other shape the library answers hint 31 with its cost model, i.e. the
    tuner would time hint 0 twice and could record 31 from noise (tests/test_tune_table.py rejects such entries)."""
    pitch = getattr(d, "pix_pitch", 0) or d.Cin
    return (int(d.mode) != 0 and d.KH == 3 and d.KW == 3 and d.stride == 1 and d.Cin == 128 and d.N == 128 and 0 < d.W <= 31
            and getattr(d, "pad_t", 1) == 1 and getattr(d, "pad_l", 1) == 1 and getattr(d, "OH", d.H) == d.H
            and getattr(d, "OW", d.W) == d.W and not d.residual and not d.out_f32 and not getattr(d, "a_scale", None)
            # ... and the rest of what tfimm_hip_gemm asks before it honours hint 31 (csrc/gemm.hip): the default stride in w, no row
            # remap, no folded LayerNorm, K-padded weights, a pixel pitch of exactly 128 channels, 16-byte aligned output rows
            and getattr(d, "stride_w", 0) in (0, d.stride) and getattr(d, "remap_in", 0) == 0 and not getattr(d, "ln_stats", None)
            and getattr(d, "ldw", d.K) >= d.K and pitch == 128 and d.ldc % 8 == 0)


# a layer with a second A operand runs on the persistent tiles only (28 = the deep-ring schedule and 30 = the duo kernel have no
# such flavour: the library answers them with another tile)
DUAL_CANDIDATES = (0, 21, 22, 23, 24, 25, 26, 27, 29)


def candidates_for(d):
    if getattr(d, "a2", None):
        return DUAL_CANDIDATES
    if getattr(d, "ln_stats", None):
        return LN_CANDIDATES
    c = SCALE_CANDIDATES if getattr(d, "a_scale", None) else CANDIDATES
    return c if strip_shape(d) else tuple(h for h in c if h != 31)


def _parse_remap(spec: str):
    """``TFIMM_TUNE_REMAP="21:30,28:30"`` -- A/B switch: table entries with the left hint are launched with the right one."""
    out = {}
    for item in filter(None, (spec or "").split(",")):
        a, b = item.split(":")
        out[int(a)] = int(b)
    return out


_REMAP = _parse_remap(os.environ.get("TFIMM_TUNE_REMAP", ""))


def lookup(d) -> int:
    h = TABLE.get(key_of(d), 0)
    if _REMAP and h in _REMAP:
        r = _REMAP[h]
        # hint 30 (two workgroups per CU) has vector epilogues only: the library falls back by itself otherwise
        return r if r in candidates_for(d) else h
    return h


def load(path: str = _PATH) -> int:
    if not os.path.exists(path):
        return 0
    with open(path) as f:
        TABLE.update({k: int(v) for k, v in json.load(f).items()})
    return len(TABLE)


def save(path: str = _PATH):
    with open(path, "w") as f:
        json.dump(dict(sorted(TABLE.items())), f, indent=0)


load()
