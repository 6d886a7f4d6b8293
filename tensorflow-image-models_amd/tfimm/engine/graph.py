"""Layer-program builder and executor.

A model's ``lower()`` method traces its forward pass into a flat list of fused-kernel
invocations (a *program*) using :class:`Builder`; each builder method corresponds to one
C-ABI call of libtfimm_hip.so and cites the reference op sequence it fuses.  The program is
pure host data (shapes, packed numpy weights) until :meth:`Program.upload`; a
:class:`Plan` binds it to device buffers for one batch size and runs it by calling the C
ABI through ctypes -- there is no other execution path.

Activations are bf16 ``[B * rows_per_image, C]`` row-major (NHWC flattened); a tensor may
carry its spatial extent ``(H, W)``.
"""
import ctypes as C
import os
import zlib
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from ..utils.etc import same_padding
from . import pack, precision, tune


# ---------------------------------------------------------------------------------------
# symbolic tensors / constants
# ---------------------------------------------------------------------------------------
@dataclass
class TRef:
    """Activation tensor of the program: per-image ``rows x C`` (x batch at run time)."""
    id: int
    rows: int              # rows per image (H*W, tokens, or 1)
    C: int
    H: int = 0
    W: int = 0
    dtype: str = "bf16"    # "bf16" | "f32"
    keep: bool = False     # never recycle the buffer (outputs / returned features)
    # a view onto another tensor's buffer: (parent id, element offset per image-row stride)
    name: str = ""

    @property
    def itemsize(self) -> int:
        return 2 if self.dtype == "bf16" else 4

    @property
    def bytes_per_image(self) -> int:
        return self.rows * self.C * self.itemsize


@dataclass
class Const:
    """Device constant (packed weight).  ``host`` is a numpy array (uint16 = bf16 bits); it is released once the constant
    is on the device unless ``keep_host`` (constants a kernel takes by HOST pointer)."""
    id: int
    host: Optional[np.ndarray]
    name: str = ""
    nbytes: int = 0
    keep_host: bool = False
    key: Optional[tuple] = None      # (name, shape, dtype, checksum): identical constants of other programs share one upload


@dataclass
class Op:
    kind: str
    inputs: List[int]            # TRef ids read
    output: Optional[int]        # TRef id written
    consts: Dict[str, int]       # role -> Const id
    attrs: Dict[str, object]
    cite: str = ""               # reference file:line this op replaces
    extra_outputs: List[int] = field(default_factory=list)  # further tensors the kernel writes


class Program:
    def __init__(self):
        # "bf16": the product path; "fp32": the verification path (engine/precision.py) -- every activation tensor is
        # float32 and the plan binds the tfimm_hip_ref_* kernels
        self.precision = precision.get()
        self.tensors: List[TRef] = []
        self.consts: List[Const] = []
        self.ops: List[Op] = []
        self.input: Optional[TRef] = None       # raw image tensor as handed in by the caller
        self.input_shape: Tuple[int, int, int] = (0, 0, 0)
        self.outputs: Dict[str, TRef] = {}
        self._dev_consts = None
        self._dev_consts_device = None
        self.const_cache: Optional[dict] = None     # set by the owning Model

    # -- construction helpers -------------------------------------------------------------
    def new_tensor(self, rows, C, H=0, W=0, dtype="bf16", name="") -> TRef:
        if dtype == "bf16" and self.precision == "fp32":
            dtype = "f32"
        t = TRef(len(self.tensors), int(rows), int(C), int(H), int(W), dtype, False, name)
        self.tensors.append(t)
        return t

    def new_const(self, host: np.ndarray, name="", keep_host=False) -> int:
        h = np.ascontiguousarray(host)
        key = (name, h.shape, h.dtype.str, zlib.crc32(h.view(np.uint8).reshape(-1)))
        c = Const(len(self.consts), h, name, h.nbytes, keep_host, key)
        self.consts.append(c)
        return c.id

    def add(self, kind, inputs, output, consts=None, cite="", extra_outputs=(), **attrs) -> Op:
        op = Op(kind, [t.id for t in inputs], None if output is None else output.id,
                consts or {}, attrs, cite, [t.id for t in extra_outputs])
        self.ops.append(op)
        return op

    def mark_output(self, name: str, t: TRef):
        self.tensors[t.id].keep = True
        self.outputs[name] = t

    # -- statistics (used by bench.py / DESIGN.md for algorithmic work) ---------------------
    def flops_per_image(self) -> int:
        total = 0
        for op in self.ops:
            a = op.attrs
            if op.kind in ("gemm", "stem_pool"):
                total += 2 * a["M"] * a["N"] * a["K_true"]
            elif op.kind in ("attention", "talking_heads_attention", "conv_chain", "mlp_fused"):
                total += a["flops"]
            elif op.kind == "dwconv":
                total += 2 * a["OH"] * a["OW"] * a["C"] * a["k"] * a["k"]
            elif op.kind in ("grouped_conv", "expand_dwconv"):
                total += a["flops"]
        return total

    def weight_bytes(self) -> int:
        return sum(c.nbytes for c in self.consts)

    # -- buffer planning ---------------------------------------------------------------------
    def plan_buffers(self) -> Tuple[Dict[int, int], List[int]]:
        """Greedy liveness-based slab assignment.  Returns (tensor id -> slab, slab bytes
        per image).  A tensor's slab is recycled after its last reader; outputs of an op are
        placed before its inputs are released, so a kernel never reads and writes one slab."""
        last_use: Dict[int, int] = {}
        for i, op in enumerate(self.ops):
            for t in op.inputs:
                last_use[t] = i
            for t in ([op.output] if op.output is not None else []) + op.extra_outputs:
                last_use[t] = max(last_use.get(t, i), i)
        slabs: List[int] = []
        free: List[int] = []
        assign: Dict[int, int] = {}
        for i, op in enumerate(self.ops):
            outs = ([op.output] if op.output is not None else []) + op.extra_outputs
            for out in outs:
                if out in assign:
                    continue
                need = self.tensors[out].bytes_per_image
                best = None
                for s_ in free:
                    if slabs[s_] >= need and (best is None or slabs[s_] < slabs[best]):
                        best = s_
                if best is None and free:
                    best = max(free, key=lambda s_: slabs[s_])  # grow the largest free slab
                    slabs[best] = need
                if best is None:
                    slabs.append(need)
                    best = len(slabs) - 1
                else:
                    free.remove(best)
                assign[out] = best
            for t in set(op.inputs + outs):
                if last_use.get(t) == i and not self.tensors[t].keep and t in assign:
                    if assign[t] not in free:
                        free.append(assign[t])
        return assign, slabs

    # -- device side ------------------------------------------------------------------------
    def upload(self, device="cuda"):
        """Packed constants -> device.  ``self.const_cache`` (the owning model's, shared by all its programs: another
        input size, the feature-returning variant) maps a constant's key to its device tensor, so identical constants are
        uploaded once; host copies are dropped afterwards."""
        import torch
        if self._dev_consts is not None and self._dev_consts_device == device:
            return
        self._dev_consts_device = device
        cache = self.const_cache if self.const_cache is not None else {}
        dev = []
        for c in self.consts:
            ck = (device,) + c.key
            t = cache.get(ck)
            if t is None:
                h = c.host
                if h is None:
                    # the host copy went with an earlier upload (another device): copy that device tensor over
                    src = next((v for k, v in cache.items() if k[1:] == c.key), None)
                    if src is None:
                        raise RuntimeError(f"constant {c.key!r}: no host copy and no uploaded copy to take it from")
                    t = src.to(device)
                elif h.dtype == np.uint16:
                    t = torch.from_numpy(h.view(np.int16).copy()).to(device)
                else:
                    t = torch.from_numpy(h.copy()).to(device)
                cache[ck] = t
            dev.append(t)
            if not c.keep_host and device != "cpu":
                c.host = None
        self._dev_consts = dev

    def make_plan(self, batch: int, device: str = "cuda") -> "Plan":
        return Plan(self, batch, device)

    def live_across(self, op_index: int) -> List[int]:
        """Tensors written by ops [0, op_index) that ops [op_index, ...) still read (or that are program outputs)."""
        written, needed = set(), set()
        for i, op in enumerate(self.ops):
            outs = ([op.output] if op.output is not None else []) + list(op.extra_outputs)
            if i < op_index:
                written.update(outs)
            else:
                needed.update(op.inputs)
                needed.update(o for o in outs if o in written)        # ops that update an earlier tensor in place
        keep = {t.id for t in self.tensors if t.keep}
        return sorted(t for t in written if t in needed or t in keep)

    def supports_branches(self) -> bool:
        """Every program may run as parallel branches.  (Round 3 excluded the talking-heads programs -- CaiT: the H = 4 launch
        was not bit-reproducible next to GEMM launches of another stream.  Cause, found in round 4: hipcc had emitted packed fp32
        FMAs that take the HIGH half of a source pair into the LOW result (``op_sel:[0,1,0]``), which gfx950 evaluates
        wrongly in lanes 48..63 while waves of the persistent GEMM kernel share the SIMD; the kernel no longer contains such
        instructions, ``tools/isa_lint.py`` keeps them out of every product kernel, and
        ``tests/test_gpu_branches.py::test_talking_heads_next_to_the_gemm_that_disturbed_it`` launches the pair 60 times.
        profiles/NOTES_r04.md section 1.)"""
        return True

    def make_branches(self, batch: int, parts: int = 2, device: str = "cuda") -> List["Plan"]:
        """``parts`` plans over consecutive slices of one batch (sizes as even as possible, each with its own activation
        slabs, all on this program's weights): the branches of ``CapturedBranches``."""
        return [Plan(self, nb, device) for nb in branch_sizes(batch, parts)]


def branch_sizes(batch: int, parts: int) -> List[int]:
    """Images per branch: ``parts`` consecutive slices of the batch, as even as possible, none empty."""
    parts = max(1, min(int(parts), int(batch)))
    return [batch // parts + (1 if i < batch % parts else 0) for i in range(parts)]


# ---------------------------------------------------------------------------------------
# Builder: one method per fused kernel
# ---------------------------------------------------------------------------------------
class Builder:
    def __init__(self, weights: Dict[str, np.ndarray]):
        self.w = weights
        self._orig_w = weights
        self.p = Program()
        self.fp32 = self.p.precision == "fp32"      # verification path: no cross-layer fusion, unrounded weights

    # -- weights ---------------------------------------------------------------------------
    def wget(self, name: str) -> np.ndarray:
        if name not in self.w:
            raise KeyError(f"missing weight '{name}'")
        return np.asarray(self.w[name], dtype=np.float32)

    def define(self, name: str, value: np.ndarray) -> str:
        """Register a weight derived on the host (e.g. the k and v kernels of CaiT's ClassAttention
        side by side, the q kernel with the attention scale folded in) under a new name."""
        if self.w is self._orig_w:
            self.w = dict(self._orig_w)
        self.w[name] = np.asarray(value, dtype=np.float32)
        return name

    def act_const(self, value: np.ndarray, name: str) -> int:
        """A constant in ACTIVATION storage (e.g. position embeddings added as a GEMM residual): bf16 bits on the product
        path, float32 on the verification path."""
        v = np.ascontiguousarray(value, dtype=np.float32)
        return self.p.new_const(v if self.fp32 else pack.to_bf16_bits(v), name)

    def bn(self, prefix: str, eps: float):
        """Folded inference BatchNorm -> (scale, shift).  Keras BN: layers/factory.py:22-37."""
        return pack.bn_scale_shift(self.wget(prefix + "/gamma"), self.wget(prefix + "/beta"),
                                   self.wget(prefix + "/moving_mean"),
                                   self.wget(prefix + "/moving_variance"), eps)

    # -- input -----------------------------------------------------------------------------
    def image_input(self, H: int, W: int, cin: int) -> TRef:
        """Caller's NHWC float image -> bf16 NHWC with padded channels (tfimm_hip_cast_input)."""
        p = self.p
        cpad = cin if self.fp32 else pack.pad_channels(cin)
        raw = p.new_tensor(H * W, cin, H, W, dtype="raw", name="input")
        raw.keep = True
        p.input = raw
        p.input_shape = (H, W, cin)
        x = p.new_tensor(H * W, cpad, H, W, name="input_bf16")
        self._cast_op = p.add("cast_input", [raw], x, c_in=cin, c_out=cpad, H=H, W=W, pad=(0, 0, 0, 0),
                              cite="Keras input autocast")
        self._cast_out = x.id
        return x

    # -- convolution / dense -----------------------------------------------------------------
    def conv(self, x: TRef, kernel: str, *, stride=1, padding=0, bn: Optional[str] = None,
             bn_eps=1e-5, bias: Optional[str] = None, act="", residual: Optional[TRef] = None,
             act_after_res=False, a_scale: Optional[TRef] = None, flatten=False,
             remap=None, res_const=None, res_mod=0, then_maxpool=None, flops_k: Optional[int] = None,
             fold_shortcut=None, cite="", name="") -> TRef:
        """Conv2D (+ZeroPadding2D / "same") + folded BN / bias + activation + residual.

        ``fold_shortcut=(x2, kernel2, bn2, stride2)``: this is the last 1x1 convolution of a residual block whose shortcut
        is ``bn2(conv1x1_stride2(x2))`` (resnet.py:282-290, 315-330): instead of a launch that writes the shortcut tensor and
        a ``residual`` operand that reads it back, the shortcut's input channels become further k-tiles of THIS GEMM
        (tfimm_gemm_desc::a2; ``can_fold_shortcut`` says when: channel counts of both operands and of the output multiples of 8) -- ``act`` is then the
        block's final activation.

        ``padding``: int (symmetric, as the reference's ZeroPadding2D + VALID), "same"
        (TF asymmetric, layers/conv.py:61) or an explicit ``((top, bottom), (left, right))``.
        ``then_maxpool=(k, stride, pad)``: the max pooling that follows (resnet.py:538-540).  The ResNet stem shape
        (7x7 stride 2 on the RGB image, 64 channels, ReLU, 3x3 / 2 / 1 pooling, at most 112 output columns) runs as ONE
        kernel that never writes the convolution output (tfimm_hip_stem_conv_pool); anything else is the convolution
        followed by ``maxpool``.
        """
        p = self.p
        k = self.wget(kernel)
        kh, kw, cin, cout = k.shape
        assert x.H > 0 and x.W > 0, "conv needs a spatial tensor"
        if isinstance(padding, str):
            if padding == "same":
                OH, pt, _ = same_padding(x.H, kh, stride)
                OW, pl, _ = same_padding(x.W, kw, stride)
            elif padding == "valid":
                pt = pl = 0
                OH = (x.H - kh) // stride + 1
                OW = (x.W - kw) // stride + 1
            else:
                raise ValueError(padding)
        else:
            if isinstance(padding, int):
                pt = pb = pl = pr = padding
            else:
                (pt, pb), (pl, pr) = padding
            OH = (x.H + pt + pb - kh) // stride + 1
            OW = (x.W + pl + pr - kw) // stride + 1
        scale = shift = None
        if bn is not None:
            scale, shift = self.bn(bn, bn_eps)
        if bias is not None:
            b = self.wget(bias)
            shift = b if shift is None else shift + b * scale
        M = OH * OW
        out = p.new_tensor(M, cout, 0 if flatten else OH, 0 if flatten else OW, name=name or kernel)
        consts = {}
        # K_true feeds the algorithmic FLOP count; ``flops_k`` overrides it where the kernel handed in is an expansion
        # of the reference's (grouped 3x3 as block-diagonal, pooling folded into the taps)
        attrs = dict(M=M, N=cout, K_true=flops_k or kh * kw * cin, act=act, act_after_res=act_after_res,
                     out_f32=0, res_mod=res_mod, remap=remap)
        pointwise = (kh == 1 and kw == 1 and stride == 1 and pt == 0 and pl == 0 and x.C == cin)
        # RGB stem / patch embedding with an even stride: the zero padding is written once by the input
        # cast (tfimm_hip_cast_input_pad) and the conv runs on the pixel-PAIR view [Hp][Wp/2][8] of the
        # padded 4-channel image -- no bounds checks, Cin % 8 == 0, so the operand tiles go by LDS-DMA.
        pair_view = (x.C == 4 and getattr(self, "_cast_out", None) == x.id and stride % 2 == 0 and a_scale is None
                     and self._cast_op.attrs["pad"] == (0, 0, 0, 0) and not self._cast_op.attrs.get("used")
                     and not self.fp32)
        if pair_view:
            kwp = (kw + 1) // 2 * 2
            wp = max(x.W + pl, (OW - 1) * stride + kwp)
            wp += wp & 1
            hp = max(x.H + pt, (OH - 1) * stride + kh)
            self._cast_op.attrs.update(pad=(pt, hp - x.H - pt, pl, wp - x.W - pl), used=True)
            xt = p.tensors[x.id]
            xt.rows, xt.H, xt.W = hp * wp, hp, wp
            wt, bvec, kk, _ = pack.pack_conv(k, scale, shift, 4)   # K order (ky, kx < kwp, c < 4) == (ky, pair, 8)
            if (then_maxpool == (3, 2, 1) and kh == 7 and kw == 7 and stride == 2 and cout == 64 and act == "relu"
                    and OW <= 112 and wp // 2 <= 116 and residual is None and remap is None and not flatten
                    and bvec is not None and os.environ.get("TFIMM_NO_STEM_FUSION", "0") != "1"):
                PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
                pooled = p.new_tensor(PH * PW, cout, PH, PW, name=name or kernel)
                consts = {"wt": p.new_const(wt, kernel), "bias": p.new_const(bvec, kernel + ":bias")}
                p.add("stem_pool", [x], pooled, consts, cite=cite, Hp=hp, Wp2=wp // 2, OH=OH, OW=OW, ldw=wt.shape[1],
                      N=cout, K_true=kh * kw * cin, M=M)
                return pooled
            attrs.update(mode=1, K=kk, H=hp, W=wp // 2, Cin=8, KH=kh, KW=kwp // 2, stride=stride, stride_w=stride // 2,
                         pad_t=0, pad_l=0, OH=OH, OW=OW)
        elif pointwise:
            wt, bvec = pack.pack_dense(k.reshape(cin, cout) * (1.0 if scale is None else scale.reshape(1, cout)), shift,
                                       fp32=self.fp32)
            attrs.update(mode=0, K=cin, lda=x.C, a_rows_per_image=x.rows)
            if fold_shortcut is not None:
                assert residual is None and a_scale is None and remap is None
                wt2, shift2, du = self._shortcut_operand(fold_shortcut, cout, cin, OH, OW, bn_eps)
                wt = np.concatenate([wt, wt2], axis=1)          # [N][ceil64(K) + taps * ceil64(K2)]: the second operand's k-tiles follow
                bvec = shift2 if bvec is None else bvec + shift2
                attrs["dual"] = du
                attrs["K_true"] = cin + du["K2"]
        else:
            assert a_scale is None
            wt, bvec, kk, mode = pack.pack_conv(k, scale, shift, x.C, fp32=self.fp32)
            attrs.update(mode=mode, K=kk, H=x.H, W=x.W, Cin=x.C, KH=kh, KW=kw, stride=stride,
                         pad_t=pt, pad_l=pl, OH=OH, OW=OW)
            if fold_shortcut is not None:          # the 3x3 conv2 of a basic block + its block's 1x1 shortcut convolution
                assert mode == 1 and residual is None and remap is None
                wt2, shift2, du = self._shortcut_operand(fold_shortcut, cout, x.C, OH, OW, bn_eps)
                wt = np.concatenate([wt, wt2], axis=1)
                bvec = shift2 if bvec is None else bvec + shift2
                attrs["dual"] = du
                attrs["K_true"] = kh * kw * cin + du["K2"]
        consts["wt"] = p.new_const(wt, kernel)
        attrs["ldw"] = wt.shape[1]
        if bvec is not None:
            consts["bias"] = p.new_const(bvec, kernel + ":bias")
        if res_const is not None:
            consts["residual"] = res_const
        ins = [x]
        if residual is not None:
            assert residual.C == cout and residual.rows == M
            ins.append(residual)
            attrs["has_residual"] = True
            attrs["ldr"] = residual.C
        if a_scale is not None:
            assert a_scale.C == cin and pointwise
            ins.append(a_scale)
            attrs["has_scale"] = True
        if remap is not None:
            # rows land in a larger token buffer: (rows_in_per_image, rows_out_per_image, offset)
            out.rows = remap[1]
        if attrs.get("dual"):
            ins.append(fold_shortcut[0])
        else:
            assert fold_shortcut is None, "fold_shortcut: a 1x1 convolution or a Cin % 8 == 0 gather (see can_fold_shortcut)"
        p.add("gemm", ins, out, consts, cite=cite, **attrs)
        if then_maxpool is not None:
            return self.maxpool(out, *then_maxpool, cite=cite)
        return out

    def _shortcut_operand(self, spec, cout: int, cin: int, OH: int, OW: int, bn_eps: float):
        """Weights [N][taps * ceil64(K2)], folded-BN shift and descriptor fields of a shortcut convolution taken as the second A
        operand: ``spec = (x2, kernel2, bn2, stride2[, window])`` -- a 1x1 / stride-s convolution + BatchNorm (resnet.py:315-330), or,
        ``window`` = 2, ResNet-D's AveragePooling2D(2, 2) + 1x1 convolution + BatchNorm (resnet.py:295-312) as the 2x2 / stride-2
        convolution whose four taps are the 1x1 kernel / 4 (exact in bf16: a power of two)."""
        x2, kernel2, bn2, stride2 = spec[:4]
        window = spec[4] if len(spec) > 4 else 1
        assert self.can_fold_shortcut(x2, stride2, cout, cin)
        k2 = self.wget(kernel2)
        assert k2.shape[:2] == (1, 1) and k2.shape[2] == x2.C and k2.shape[3] == cout
        assert ((x2.H - window) // stride2 + 1, (x2.W - window) // stride2 + 1) == (OH, OW), (x2.H, x2.W, window, stride2, OH, OW)
        scale2, shift2 = self.bn(bn2, bn_eps)
        tap = k2.reshape(x2.C, cout) * scale2.reshape(1, cout) / float(window * window)
        wt2 = np.concatenate([pack.pack_dense(tap, None)[0]] * (window * window), axis=1)
        du = dict(K2=x2.C, lda2=x2.C, stride=stride2, H=x2.H, W=x2.W, OH=OH, OW=OW, window=window if window > 1 else 0)
        return wt2, shift2, du

    def can_fold_shortcut(self, x2: TRef, stride2: int, cout: int, cin: int = 8) -> bool:
        """Whether ``conv(..., fold_shortcut=(x2, ...))`` can take a 1x1 / stride-``stride2`` shortcut convolution of ``x2`` as a
        second A operand: the bf16 product path on a library that has the entry (ABI >= 4), 16-byte aligned channel runs on
        both sides.  TFIMM_NO_FOLD_SHORTCUT=1 keeps the shortcut a launch of its own (A/B)."""
        from . import ffi
        return (not self.fp32 and ffi.ABI >= 4 and x2.H > 0 and x2.W > 0 and x2.C % 8 == 0 and cout % 8 == 0 and cin % 8 == 0 and stride2 >= 1
                and os.environ.get("TFIMM_NO_FOLD_SHORTCUT", "0") != "1")

    def conv_chain(self, x: TRef, kernel1: str, bn1: str, kernel2: str, bn2: str, *, stride=1, padding=1, bn_eps=1e-5,
                   act1="relu", act2="relu", residual: Optional[TRef] = None, shortcut_conv=None, cite="") -> Optional[TRef]:
        """k x k Conv2D + BN + act1 followed by a 1x1 Conv2D + BN (+ residual) + act2 as ONE launch
        (tfimm_hip_conv_chain): the tail of a ResNet bottleneck block with the intermediate kept in LDS.  Returns None
        when the shape is outside what that kernel is built for (the caller then lowers the two convolutions)."""
        p = self.p
        k1, k2 = self.wget(kernel1), self.wget(kernel2)
        kh, kw, cin, c1 = k1.shape
        n2 = k2.shape[3]
        # what tfimm_hip_conv_chain is built for: 3x3 / stride 1 / pad 1 over 64 -> 64 channels, rows of at most 63 pixels
        # (gemm_chain_kernel.h), or over 128 -> 128 channels, rows of at most 31 pixels (conv_strip.hip; no shortcut convolution)
        shape_ok = ((kh, kw, cin, c1) == (3, 3, 64, 64) and x.W <= 63) or \
                   ((kh, kw, cin, c1) == (3, 3, 128, 128) and x.W <= 31 and shortcut_conv is None
                    and os.environ.get("TFIMM_NO_CHAIN128", "0") != "1")
        if (x.C != cin or not shape_ok or stride != 1 or int(padding) != 1
                or n2 not in (256, 512) or k2.shape[:3] != (1, 1, c1) or os.environ.get("TFIMM_NO_CHAIN", "0") == "1"
                or self.fp32):
            return None
        if shortcut_conv is not None:
            x0, kds, _ = shortcut_conv
            if (residual is not None or self.wget(kds).shape != (1, 1, 64, n2) or x0.C != 64 or x0.rows != x.rows
                    or (act1, act2) != ("relu", "relu")):
                return None
        pad = int(padding)
        OH = (x.H + 2 * pad - kh) // stride + 1
        OW = (x.W + 2 * pad - kw) // stride + 1
        s1, t1 = self.bn(bn1, bn_eps)
        s2, t2 = self.bn(bn2, bn_eps)
        wt1, b1, kk, mode = pack.pack_conv(k1, s1, t1, x.C)
        assert mode == 1 and kk == kh * kw * cin and wt1.shape[1] == kk
        # GEMM 2 reads its A operand straight from GEMM 1's accumulator registers: W2's K axis goes in that order
        wt2, b2 = pack.pack_dense((k2.reshape(c1, n2) * s2.reshape(1, n2))[pack.chain_k_order(c1)], t2)
        out = p.new_tensor(OH * OW, n2, OH, OW, name=kernel2)
        consts = {"w1": p.new_const(wt1, kernel1), "b1": p.new_const(b1, kernel1 + ":bias"),
                  "w2": p.new_const(wt2, kernel2)}
        flops = 2 * OH * OW * (c1 * kh * kw * cin + n2 * c1)
        ds = None
        if shortcut_conv is not None:
            # ``shortcut_conv = (block input, kernel, bn)``: a 1x1 / stride 1 shortcut convolution of a 64-channel block input
            # (first block of a stage) multiplied inside this launch; its folded-BN shift joins b2
            x0, kds, bnds = shortcut_conv
            wds = self.wget(kds)
            sds, tds = self.bn(bnds, bn_eps)
            b2 = (b2 + tds).astype(np.float32)
            consts["ds_w"] = p.new_const(pack.pack_chain_ds(wds.reshape(64, n2) * sds.reshape(1, n2)), kds + ":frag")
            flops += 2 * OH * OW * 64 * n2
            ds = x0
        consts["b2"] = p.new_const(b2, kernel2 + (":bias+shortcut" if ds is not None else ":bias"))
        ins = [x] + ([residual] if residual is not None else []) + ([ds] if ds is not None else [])
        if residual is not None:
            assert residual.rows == OH * OW and residual.C == n2
        p.add("conv_chain", ins, out, consts, cite=cite, H=x.H, W=x.W, Cin=cin, KH=kh, KW=kw, stride=stride, pad=pad,
              OH=OH, OW=OW, C1=c1, N2=n2, ldw1=wt1.shape[1], ldw2=wt2.shape[1], act1=act1, act2=act2,
              has_residual=residual is not None, has_ds=ds is not None, flops=flops)
        return out

    def grouped_conv3x3(self, x: TRef, kernel: str, groups: int, *, stride=1, bn: Optional[str] = None, bn_eps=1e-5,
                        act="", cite="") -> Optional[TRef]:
        """ZeroPadding2D(1) + Conv2D(3x3, stride, groups) + folded BN + activation on 32-channel super-groups
        (tfimm_hip_grouped_conv3x3).  None when the group width does not fit that kernel (> 32 channels per group, or
        32 not a multiple of it): the caller then falls back to the dense block-diagonal expansion."""
        p = self.p
        k = self.wget(kernel)
        kh, kw, w, c = k.shape
        if ((kh, kw) != (3, 3) or c != w * groups or x.C != c or w > 32 or 32 % w or c % 32 or stride not in (1, 2)
                or os.environ.get("TFIMM_NO_GROUPED", "0") == "1" or self.fp32):
            return None
        scale = shift = None
        if bn is not None:
            scale, shift = self.bn(bn, bn_eps)
        wfrag = pack.pack_grouped3x3(k, groups, scale)
        bvec = np.zeros(c, np.float32) if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
        OH, OW = (x.H + 2 - 3) // stride + 1, (x.W + 2 - 3) // stride + 1
        out = p.new_tensor(OH * OW, c, OH, OW, name=kernel)
        consts = {"w": p.new_const(wfrag, kernel), "bias": p.new_const(bvec, kernel + ":bias")}
        p.add("grouped_conv", [x], out, consts, cite=cite, H=x.H, W=x.W, C=c, stride=stride, act=act,
              flops=2 * OH * OW * c * 9 * w)
        return out

    def grouped_conv_split(self, x: TRef, kernel: str, groups: int, *, stride=1, padding=1, bn: Optional[str] = None,
                           bn_eps=1e-5, act="", cite="") -> Optional[TRef]:
        """Conv2D(groups) with wide groups as ONE tfimm_hip_gemm launch PER GROUP: the implicit-GEMM gather reads the
        group's channel slice in place (``pix_pitch`` = the tensor's channel count, ``a`` offset to the slice) and writes its
        output channels into their slice of the result (``ldc``, ``out_col``) -- no block-diagonal expansion, no channel
        shuffle passes.  Needs group widths that are multiples of 8 (16-byte slices)."""
        p = self.p
        k = self.wget(kernel)
        kh, kw, w, c = k.shape
        if c % groups or x.C != w * groups or ((w % 8 or (c // groups) % 8) and not self.fp32):
            return None
        wo = c // groups
        pad = int(padding)
        OH = (x.H + 2 * pad - kh) // stride + 1
        OW = (x.W + 2 * pad - kw) // stride + 1
        scale = shift = None
        if bn is not None:
            scale, shift = self.bn(bn, bn_eps)
        out = p.new_tensor(OH * OW, c, OH, OW, name=kernel)
        for g in range(groups):
            sl = slice(g * wo, (g + 1) * wo)
            wt, bvec, kk, mode = pack.pack_conv(k[..., sl], None if scale is None else scale[sl],
                                                None if shift is None else shift[sl], w, fp32=self.fp32)
            consts = {"wt": p.new_const(wt, f"{kernel}:g{g}")}
            if bvec is not None:
                consts["bias"] = p.new_const(bvec, f"{kernel}:g{g}:bias")
            p.add("gemm", [x], out, consts, cite=cite, M=OH * OW, N=wo, K=kk, K_true=kk, mode=mode, H=x.H, W=x.W, Cin=w,
                  KH=kh, KW=kw, stride=stride, pad_t=pad, pad_l=pad, OH=OH, OW=OW, ldw=wt.shape[1], act=act,
                  act_after_res=False, out_f32=0, pix_pitch=x.C, a_byte_offset=g * w * x.itemsize, out_col=g * wo, ldc=c)
        return out

    def dense(self, x: TRef, kernel: str, bias: Optional[str] = None, *, act="",
              residual: Optional[TRef] = None, out_f32=False, row_select: Optional[Tuple[int, int]] = None,
              in_cols: Optional[Tuple[int, int]] = None,
              out: Optional[TRef] = None, out_col: int = 0, out_scale: Optional[str] = None,
              residual_row: Optional[int] = None, out_row: Optional[int] = None,
              ln: Optional[Tuple[str, float, TRef]] = None, cite="", name="") -> TRef:
        """tf.keras.layers.Dense (+ activation, + residual add).

        ``ln=(prefix, eps, stats)``: ``x`` is the RAW input of LayerNormalization ``prefix`` whose output this layer reads;
        the normalisation is folded into the layer (gamma into the weights, beta . W into the bias, per-row mean / rstd
        from ``stats`` = ``row_stats(x, eps)`` applied in the GEMM epilogue) and the normalised tensor never exists.

        ``row_select=(first_row, count)`` applies the layer to ``count`` rows per image
        starting at ``first_row`` (e.g. the class token x[:, 0], vit.py:462) without a copy.
        ``out``/``out_col``: write into columns [out_col, out_col+N) of an existing tensor
        (tf.stack of the two DeiT heads, vit.py:474-476).
        ``residual_row`` / ``out_row``: with one row per image, take the residual from / write the result
        to that token row of a multi-row tensor (CaiT's class-token-only blocks update x[:, 0] of the
        token tensor in place, cait.py:186-200).
        """
        p = self.p
        k = self.wget(kernel)
        if k.ndim == 4:     # 1x1 Conv2D used as a Dense (ConvMLP, layers/transformers.py:238-252)
            assert k.shape[0] == 1 and k.shape[1] == 1
            k = k[0, 0]
        kin, kout = k.shape
        bvec_in = None if bias is None else self.wget(bias)
        if out_scale is not None:
            # per-output-channel scale behind the layer (LayerScale, convnext.py:228) folded into it:
            # g * (x W + b) = x (W g) + b g
            g = self.wget(out_scale).reshape(1, kout)
            k = k * g
            if bvec_in is not None:
                bvec_in = bvec_in * g.reshape(kout)
        if in_cols is None:
            assert kin == x.C, f"{kernel}: in={kin} but tensor has C={x.C}"
        else:
            assert in_cols[1] == kin and in_cols[0] + kin <= x.C
        if ln is not None:
            assert residual is None and row_select is None and in_cols is None and not out_f32 and out is None
            assert kin % 8 == 0 and kout % 8 == 0
            gam, bet = self.wget(ln[0] + "/gamma").astype(np.float64), self.wget(ln[0] + "/beta").astype(np.float64)
            shift = bet @ k.astype(np.float64)                     # beta . W (+ b): what LN's beta contributes to every row
            bvec_in = (shift if bvec_in is None else shift + bvec_in).astype(np.float32)
            k = (k.astype(np.float64) * gam.reshape(kin, 1)).astype(np.float32)
        wt, bvec = pack.pack_dense(k, bvec_in, fp32=self.fp32)
        rows = x.rows
        attrs = dict(M=rows, N=kout, K=kin, K_true=kin, mode=0, lda=x.C, act=act, act_after_res=False,
                     out_f32=1 if out_f32 else 0, res_mod=0, remap=None, ldw=wt.shape[1],
                     a_rows_per_image=x.rows)
        if row_select is not None:
            first, count = row_select
            assert count == 1, "row_select takes one row per image"
            attrs.update(M=1, a_byte_offset=first * x.C * x.itemsize, lda=x.rows * x.C)
            rows = 1
        if in_cols is not None:
            attrs["a_byte_offset"] = attrs.get("a_byte_offset", 0) + in_cols[0] * x.itemsize
        if out is None:
            sp = (x.H, x.W) if (row_select is None and x.H * x.W == rows) else (0, 0)   # Dense keeps the spatial grid
            out = p.new_tensor(rows, kout, sp[0], sp[1], dtype="f32" if out_f32 else "bf16", name=name or kernel)
        else:
            attrs["out_col"] = out_col
        attrs["ldc"] = out.C
        if out_row is not None:
            assert rows == 1 and out.C == kout and out_row < out.rows
            attrs["ldc"] = out.rows * out.C
            attrs["out_byte_offset"] = out_row * out.C * out.itemsize
        consts = {"wt": p.new_const(wt, kernel + (":ln" if ln is not None else ""))}
        if bvec is not None:
            consts["bias"] = p.new_const(bvec, kernel + (":ln:bias" if ln is not None else ":bias"))
        ins = [x]
        if ln is not None:
            consts["ln_c1"] = p.new_const(pack.pack_ln_c1(wt, kout, kin), kernel + ":ln_c1")
            assert ln[2].rows == x.rows and ln[2].C == 2
            ins.append(ln[2])
            attrs["ln"] = True
        if residual is not None:
            assert residual.C == kout
            ins.append(residual)
            attrs["has_residual"] = True
            attrs["ldr"] = residual.C
            if residual_row is not None:
                assert rows == 1 and residual_row < residual.rows
                attrs["ldr"] = residual.rows * residual.C
                attrs["res_byte_offset"] = residual_row * residual.C * residual.itemsize
            else:
                assert residual.rows == rows
        p.add("gemm", ins, out, consts, cite=cite, **attrs)
        return out

    def empty(self, rows: int, C: int, dtype="bf16", name="") -> TRef:
        return self.p.new_tensor(rows, C, dtype=dtype, name=name)

    # -- normalisation -----------------------------------------------------------------------
    def can_fold_ln(self, x: TRef) -> bool:
        """LayerNormalization over x's channels can be folded into the Dense layers that read it (dense(..., ln=...))."""
        # the fold lives in the persistent LDS-DMA GEMM family only (tfimm_hip_gemm refuses ln_stats elsewhere)
        if any(os.environ.get(k, "0") == "1" for k in ("TFIMM_NO_LN_FOLD", "TFIMM_GEMM_NO_STREAM", "TFIMM_GEMM_NO_DMA")):
            return False
        return x.C % 8 == 0 and x.C <= 2048 and not self.fp32

    def ln_dense(self, x: TRef, ln_prefix: str, eps: float, kernel: str, bias: Optional[str] = None, *, act="",
                 cite_ln="", cite="") -> TRef:
        """LayerNormalization(ln_prefix) followed by a Dense layer that is its only reader: folded into one GEMM over the raw
        rows + a statistics pass when the row width allows (``can_fold_ln``), the two launches otherwise."""
        kshape = self.wget(kernel).shape
        if self.can_fold_ln(x) and kshape[-1] % 8 == 0:      # the folded epilogue stores whole 16-byte groups
            return self.dense(x, kernel, bias, act=act, ln=(ln_prefix, eps, self.row_stats(x, eps, cite=cite_ln)),
                              cite=(cite_ln + ", " + cite) if cite_ln else cite)
        return self.dense(self.layernorm(x, ln_prefix, eps, cite=cite_ln), kernel, bias, act=act, cite=cite)

    def mlp_fused(self, x: TRef, ln_prefix: str, eps: float, fc1: str, fc2: str, *, act="gelu", residual: Optional[TRef] = None,
                  out_scale: Optional[str] = None, cite="") -> Optional[TRef]:
        """residual + [out_scale *] fc2(act(fc1(LayerNormalization(x)))) as ONE launch (tfimm_hip_mlp_fused: the 4C-wide hidden
        tensor stays in registers).  ``fc1`` / ``fc2`` are layer prefixes (kernel + bias).  Returns None when the shape is
        not one the kernel is built for (C = 128, hidden = 512) -- the caller lowers the two GEMMs."""
        if self.fp32 or os.environ.get("TFIMM_NO_MLP_FUSION"):
            return None
        k1, k2 = self.wget(fc1 + "/kernel"), self.wget(fc2 + "/kernel")
        if k1.ndim == 4:     # ConvMLP: 1x1 convolutions (layers/transformers.py:238-252)
            k1, k2 = k1[0, 0], k2[0, 0]
        c, hid = k1.shape
        if x.C != c or c != 128 or hid != 4 * c or k2.shape != (hid, c) or x.itemsize != 2:
            return None
        residual = x if residual is None else residual
        assert residual.C == c and residual.rows == x.rows
        p = self.p
        gam, bet = self.wget(ln_prefix + "/gamma"), self.wget(ln_prefix + "/beta")
        w1, b1, w2, b2 = pack.pack_mlp_fused(k1, self.wget(fc1 + "/bias"), gam, bet, k2, self.wget(fc2 + "/bias"),
                                                 None if out_scale is None else self.wget(out_scale))
        consts = {"w1": p.new_const(w1, fc1 + ":mlp_w1"), "b1": p.new_const(b1, fc1 + ":mlp_b1"), "w2": p.new_const(w2, fc2 + ":mlp_w2"),
                  "b2": p.new_const(b2, fc2 + ":mlp_b2")}
        out = p.new_tensor(x.rows, c, x.H, x.W, name=fc2 + ":mlp")
        p.add("mlp_fused", [x, residual], out, consts, cite=cite, rows=x.rows, C=c, hidden=hid, act=act, eps=float(eps),
              flops=4 * x.rows * c * hid)
        return out

    def row_stats(self, x: TRef, eps: float, cite="") -> TRef:
        """(mean, rstd) of every row of x -- the statistics of a LayerNormalization that is folded into its consumers."""
        st = self.p.new_tensor(x.rows, 2, dtype="f32", name="ln_stats")
        self.p.add("row_stats", [x], st, cite=cite, rows=x.rows, d=x.C, eps=float(eps))
        return st

    def layernorm(self, x: TRef, prefix: str, eps: float, *, row_select=None, out: Optional[TRef] = None,
                  out_col: int = 0, cite="", name="") -> TRef:
        """LayerNormalization over the channel axis.  ``row_select=(row, 1)`` normalises one
        token row per image; ``out``/``out_col`` write into a column slice of an existing
        tensor (stacking the two DeiT token features without a copy)."""
        p = self.p
        g, b = self.wget(prefix + "/gamma"), self.wget(prefix + "/beta")
        assert g.shape[0] == x.C
        rows = x.rows if row_select is None else row_select[1]
        if out is None:
            out = p.new_tensor(rows, x.C, x.H if row_select is None else 0, x.W if row_select is None else 0,
                               name=name or prefix)
        else:
            assert out.rows == rows and out_col + x.C <= out.C
        consts = {"gamma": p.new_const(g, prefix + "/gamma"), "beta": p.new_const(b, prefix + "/beta")}
        if row_select is None:
            attrs = dict(rows=x.rows, x_stride=x.C, x_byte_offset=0)
        else:
            first, count = row_select
            assert count == 1
            attrs = dict(rows=1, x_stride=x.rows * x.C, x_byte_offset=first * x.C * x.itemsize)
        p.add("layernorm", [x], out, consts, cite=cite, eps=float(eps), d=x.C, y_stride=out.C,
              y_byte_offset=out_col * out.itemsize, **attrs)
        return out

    # -- attention ---------------------------------------------------------------------------
    def attention(self, qkv: TRef, heads: int, scale: float, *, window=0, shift=0, res=(0, 0),
                  rel_bias: Optional[np.ndarray] = None, cite="", name="") -> TRef:
        p = self.p
        d = qkv.C // 3
        hd = d // heads
        out = p.new_tensor(qkv.rows, d, qkv.H, qkv.W, name=name or "attn")
        consts = {}
        if rel_bias is not None:
            consts["rel_bias"] = p.new_const(np.ascontiguousarray(rel_bias, dtype=np.float32), name + ":rel_bias")
            if window:   # bias + shift mask, combined once on the host, per window kind
                consts["bias_log2"] = p.new_const(pack.swin_bias_tiles(np.asarray(rel_bias, np.float32), window, shift),
                                                  name + ":bias_log2")
        n = window * window if window else qkv.rows
        nseq = (qkv.rows // n)
        p.add("attention", [qkv], out, consts, cite=cite, heads=heads, hd=hd, scale=float(scale),
              window=window, shift=shift, res_h=res[0], res_w=res[1], n_tokens=qkv.rows,
              flops=4 * nseq * heads * n * n * hd)
        return out

    def attention_probs(self, qkv: TRef, heads: int, scale: float, cite="", name="") -> TRef:
        """softmax(scale * Q K^T) as an fp32 (heads * N, N) tensor per image: the ``attn`` entry of the feature
        dictionary (vit.py:160-163).  Only lowered when features are requested."""
        d = qkv.C // 3
        out = self.p.new_tensor(heads * qkv.rows, qkv.rows, dtype="f32", name=name or "attn_probs")
        self.p.add("attention_probs", [qkv], out, cite=cite, heads=heads, hd=d // heads, scale=float(scale),
                   n_tokens=qkv.rows)
        return out

    def talking_heads_attention(self, qkv: TRef, heads: int, scale: float, prefix: str, cite="", name="") -> TRef:
        """CaiT TalkingHeadAttention between its qkv and proj layers (cait.py:236-256); ``prefix`` holds
        the proj_l / proj_w Dense(H -> H) layers."""
        p = self.p
        d = qkv.C // 3
        hd = d // heads
        out = p.new_tensor(qkv.rows, d, qkv.H, qkv.W, name=name or prefix)
        consts = {}
        for role, nm in (("wl", "proj_l/kernel"), ("bl", "proj_l/bias"), ("ww", "proj_w/kernel"), ("bw", "proj_w/bias")):
            consts[role] = p.new_const(np.ascontiguousarray(self.wget(f"{prefix}/{nm}"), dtype=np.float32), f"{prefix}/{nm}",
                                       keep_host=True)
        # the same four arrays once more as ONE device constant [wl | bl | ww | bw]: what the MFMA kernel reads (tfimm_tha_desc.proj_dev)
        packed = np.concatenate([np.asarray(self.wget(f"{prefix}/{nm}"), dtype=np.float32).reshape(-1)
                                 for nm in ("proj_l/kernel", "proj_l/bias", "proj_w/kernel", "proj_w/bias")])
        consts["dev"] = p.new_const(np.ascontiguousarray(packed), f"{prefix}/proj_l+proj_w:packed")
        n = qkv.rows
        p.add("talking_heads_attention", [qkv], out, consts, cite=cite, heads=heads, hd=hd, scale=float(scale),
              n_tokens=n, flops=4 * heads * n * n * hd + 4 * heads * heads * n * n)
        return out

    def class_attention(self, q: TRef, kv: TRef, heads: int, cite="", name="") -> TRef:
        """CaiT ClassAttention core (cait.py:137-143): q one (pre-scaled) row per image, kv [tokens][2D]."""
        d = q.C
        assert q.rows == 1 and kv.C == 2 * d
        out = self.p.new_tensor(1, d, name=name or "class_attn")
        self.p.add("class_attention", [q, kv], out, cite=cite, heads=heads, hd=d // heads, n_tokens=kv.rows)
        return out

    def copy_rows(self, src: TRef, dst: TRef, dst_row0: int, cite="") -> None:
        """dst[:, dst_row0 : dst_row0 + src.rows] = src (one operand of a tf.concat on the token axis)."""
        assert src.C == dst.C and dst_row0 + src.rows <= dst.rows
        self.p.add("copy_rows", [src, dst], dst, cite=cite, src_rows=src.rows, dst_rows=dst.rows, dst_row0=dst_row0,
                   d=src.C)

    # -- pooling / misc ------------------------------------------------------------------------
    def maxpool(self, x: TRef, k: int, stride: int, pad: int, cite="") -> TRef:
        OH = (x.H + 2 * pad - k) // stride + 1
        OW = (x.W + 2 * pad - k) // stride + 1
        out = self.p.new_tensor(OH * OW, x.C, OH, OW, name="maxpool")
        self.p.add("maxpool", [x], out, cite=cite, H=x.H, W=x.W, C=x.C, k=k, stride=stride, pad=pad, OH=OH, OW=OW)
        return out

    def mean_rows(self, x: TRef, out_f32=False, cite="", name="") -> TRef:
        out = self.p.new_tensor(1, x.C, dtype="f32" if out_f32 else "bf16", name=name or "mean")
        self.p.add("mean_rows", [x], out, cite=cite, R=x.rows, C=x.C, out_f32=1 if out_f32 else 0)
        return out

    def token_rows(self, dst: TRef, rows_host: np.ndarray, cite="") -> None:
        """Write constant rows (class / distillation token + their pos_embed) into the first
        rows of every image of ``dst`` (tfimm_hip_bcast_rows)."""
        bits = np.ascontiguousarray(rows_host, dtype=np.float32) if self.fp32 else pack.to_bf16_bits(rows_host)
        cid = self.p.new_const(bits, "token_rows")
        self.p.add("bcast_rows", [dst], dst, {"src": cid}, cite=cite, n_rows=rows_host.shape[0],
                   d=rows_host.shape[1], dst_rows=dst.rows)

    def dwconv(self, x: TRef, kernel: str, *, stride=1, padding=0, bn=None, bn_eps=1e-5, bias=None,
               act="", squeeze=False, cite="", name=""):
        """DepthwiseConv2D + folded BN / bias + activation; optionally also emits the
        per-(image, channel) sums of its output for a following SqueezeExcite."""
        p = self.p
        k = self.wget(kernel)
        kh, kw, c, _ = k.shape
        assert kh == kw and c == x.C
        if padding == "same":
            OH, pt, _ = same_padding(x.H, kh, stride)
            OW, pl, _ = same_padding(x.W, kw, stride)
        else:
            pt = pl = int(padding)
            OH = (x.H + 2 * pt - kh) // stride + 1
            OW = (x.W + 2 * pl - kw) // stride + 1
        scale = shift = None
        if bn is not None:
            scale, shift = self.bn(bn, bn_eps)
        if bias is not None:
            b = self.wget(bias)
            shift = b if shift is None else shift + b * scale
        w, bvec = pack.pack_depthwise(k, scale, shift)
        out = p.new_tensor(OH * OW, c, OH, OW, name=name or kernel)
        consts = {"w": p.new_const(w, kernel)}
        if bvec is not None:
            consts["bias"] = p.new_const(bvec, kernel + ":bias")
        sums = None
        ins = [x]
        if squeeze and self.fp32:
            # verification path: the squeeze is its own launch (the mean of the float32 output, efficientnet_blocks.py:242)
            p.add("dwconv", ins, out, consts, cite=cite, H=x.H, W=x.W, C=c, k=kh, stride=stride, pad_t=pt, pad_l=pl, OH=OH,
                  OW=OW, act=act, sums=None)
            mean = self.mean_rows(out, out_f32=True, cite="efficientnet_blocks.py:242")
            mean.is_mean = True
            return out, mean
        if squeeze:
            sums = p.new_tensor(1, 2 * c, dtype="f32", name=(name or kernel) + ":sums")     # int64 fixed point per channel
        p.add("dwconv", ins, out, consts, cite=cite, extra_outputs=[sums] if sums is not None else [],
              H=x.H, W=x.W, C=c, k=kh, stride=stride, pad_t=pt, pad_l=pl, OH=OH, OW=OW, act=act,
              sums=None if sums is None else sums.id)
        return out, sums

    def expand_dwconv(self, x: TRef, pw_kernel: str, pw_bn: str, dw_kernel: str, dw_bn: str, *, bn_eps=1e-5, stride=1,
                      padding="same", act="", squeeze=False, cite=""):
        """Pointwise expansion (1x1 Conv2D + BN + act) followed by DepthwiseConv2D + BN + act as ONE launch
        (tfimm_hip_expand_dwconv): the expanded tensor of an inverted-residual block stays in LDS.  Returns None when the
        shape is outside what that kernel is built for (the caller lowers a GEMM and a depthwise launch); otherwise
        (output, squeeze sums or None) like ``dwconv``."""
        p = self.p
        k1, kd = self.wget(pw_kernel), self.wget(dw_kernel)
        cin, c = k1.shape[2], k1.shape[3]
        kh, kw = kd.shape[:2]
        if (k1.shape[:2] != (1, 1) or x.C != cin or cin % 8 or cin > 32 or c % 2 or kd.shape[2:] != (c, 1) or kh != kw
                or (kh, stride) not in ((3, 1), (3, 2), (5, 2)) or os.environ.get("TFIMM_NO_MBCONV_FUSION", "0") == "1"
                or self.fp32 or (stride == 1 and c > 512)):        # (stride 1: the expansion bias of all chunks sits in 2 KiB of LDS)
            return None
        if padding == "same":
            OH, pt, _ = same_padding(x.H, kh, stride)
            OW, pl, _ = same_padding(x.W, kw, stride)
        else:
            pt = pl = int(padding)
            OH = (x.H + 2 * pt - kh) // stride + 1
            OW = (x.W + 2 * pl - kw) // stride + 1
        if pt >= kh or pl >= kh or OH <= 0 or OW <= 0:
            return None
        cpad = pack.ceil_to(c, 32)
        s1, t1 = self.bn(pw_bn, bn_eps)
        s2, t2 = self.bn(dw_bn, bn_eps)
        w1 = pack.pack_expand_frag(k1.reshape(cin, c).astype(np.float32) * s1.reshape(1, c), cpad)
        wd, b2 = pack.pack_depthwise(kd, s2, t2)

        def padc(a):
            out = np.zeros(a.shape[:-1] + (cpad,), np.float32)
            out[..., :c] = a
            return out
        consts = {"w1": p.new_const(w1, pw_kernel + ":frag"), "b1": p.new_const(padc(np.asarray(t1, np.float32)), pw_kernel + ":bias"),
                  "wdw": p.new_const(padc(wd), dw_kernel + ":pad"), "b2": p.new_const(padc(b2), dw_kernel + ":bias")}
        out = p.new_tensor(OH * OW, c, OH, OW, name=dw_kernel)
        sums = p.new_tensor(1, 2 * c, dtype="f32", name=dw_kernel + ":sums") if squeeze else None     # int64 fixed point
        p.add("expand_dwconv", [x], out, consts, cite=cite, extra_outputs=[sums] if sums is not None else [],
              H=x.H, W=x.W, Cin=cin, C=c, Cpad=cpad, k=kh, stride=stride, pad_t=pt, pad_l=pl, OH=OH, OW=OW, act=act,
              sums=None if sums is None else sums.id,
              flops=2 * x.H * x.W * cin * c + 2 * OH * OW * c * kh * kw)
        return out, sums

    def stem_dwconv(self, x: TRef, stem_kernel: str, stem_bn: str, dw_kernel: str, dw_bn: str, *, bn_eps=1e-5, stride=2,
                    padding="same", dw_padding="same", act="", squeeze=False, cite=""):
        """RGB stem (3x3 / stride 2 Conv2D + BN + act) followed by the first block's depthwise 3x3 / stride 1 + BN + act as ONE
        launch (tfimm_hip_expand_dwconv, stem flavour): the stem's output -- at 190 x 190 x 48 the second largest tensor of
        EfficientNet-B4 -- never reaches HBM.  ``x`` must be the (still unpadded) 4-channel output of the input cast; its zero
        border is set here the way ``conv`` does for the pixel-pair view.  None when the shape is not what that kernel is built
        for; otherwise (output, squeeze sums or None)."""
        p = self.p
        ks, kd = self.wget(stem_kernel), self.wget(dw_kernel)
        kh, kw, cin, c = ks.shape
        if (x.C != 4 or getattr(self, "_cast_out", None) != x.id or self._cast_op.attrs["pad"] != (0, 0, 0, 0)
                or self._cast_op.attrs.get("used") or (kh, kw) != (3, 3) or stride != 2 or cin > 4 or c % 2
                or kd.shape != (3, 3, c, 1) or os.environ.get("TFIMM_NO_MBCONV_FUSION", "0") == "1" or self.fp32):
            return None
        if padding == "same":
            SH, pt, _ = same_padding(x.H, 3, 2)
            SW, pl, _ = same_padding(x.W, 3, 2)
        else:
            pt = pl = int(padding)
            SH, SW = (x.H + 2 * pt - 3) // 2 + 1, (x.W + 2 * pl - 3) // 2 + 1
        if dw_padding == "same":
            OH, dpt, _ = same_padding(SH, 3, 1)
            OW, dpl, _ = same_padding(SW, 3, 1)
        else:
            dpt = dpl = int(dw_padding)
            OH, OW = SH + 2 * dpt - 2, SW + 2 * dpl - 2
        hp, wp = max(x.H + pt, (SH - 1) * 2 + 3), max(x.W + pl, (SW - 1) * 2 + 3)
        self._cast_op.attrs.update(pad=(pt, hp - x.H - pt, pl, wp - x.W - pl), used=True)
        xt = p.tensors[x.id]
        xt.rows, xt.H, xt.W = hp * wp, hp, wp
        cpad = pack.ceil_to(c, 32)
        s1, t1 = self.bn(stem_bn, bn_eps)
        s2, t2 = self.bn(dw_bn, bn_eps)
        w1 = pack.pack_stem_frag(ks.astype(np.float32) * s1.reshape(1, 1, 1, c), cpad)
        wd, b2 = pack.pack_depthwise(kd, s2, t2)

        def padc(a):
            out = np.zeros(a.shape[:-1] + (cpad,), np.float32)
            out[..., :c] = a
            return out
        consts = {"w1": p.new_const(w1, stem_kernel + ":frag"), "b1": p.new_const(padc(np.asarray(t1, np.float32)), stem_kernel + ":bias"),
                  "wdw": p.new_const(padc(wd), dw_kernel + ":pad"), "b2": p.new_const(padc(b2), dw_kernel + ":bias")}
        out = p.new_tensor(OH * OW, c, OH, OW, name=dw_kernel)
        sums = p.new_tensor(1, 2 * c, dtype="f32", name=dw_kernel + ":sums") if squeeze else None     # int64 fixed point
        p.add("expand_dwconv", [x], out, consts, cite=cite, extra_outputs=[sums] if sums is not None else [],
              H=SH, W=SW, Cin=4, C=c, Cpad=cpad, k=3, stride=1, pad_t=dpt, pad_l=dpl, OH=OH, OW=OW, act=act,
              sums=None if sums is None else sums.id, stem=1, img_h=hp, img_w=wp,
              flops=2 * SH * SW * 9 * cin * c + 2 * OH * OW * c * 9)
        return out, sums

    def se_gate(self, sums: TRef, count: int, w_reduce: str, b_reduce: str, w_expand: str, b_expand: str,
                act: str, gate_act="sigmoid", cite="") -> TRef:
        p = self.p
        w1 = self.wget(w_reduce)  # (1,1,C,rd)
        w2 = self.wget(w_expand)  # (1,1,rd,C)
        c, rd = w1.shape[2], w1.shape[3]
        consts = {
            "w1": p.new_const(np.ascontiguousarray(w1[0, 0].T), w_reduce),
            "b1": p.new_const(self.wget(b_reduce), b_reduce),
            "w2": p.new_const(np.ascontiguousarray(w2[0, 0]), w_expand),     # [rd][C] as stored
            "b2": p.new_const(self.wget(b_expand), b_expand),
        }
        gate = p.new_tensor(1, c, dtype="f32", name="se_gate")
        # sums straight from a depthwise producer are int64 fixed point (2 floats of storage per channel); means are fp32
        fixed = sums.C == 2 * c
        assert fixed or sums.C == c
        if getattr(sums, "is_mean", False):      # fp32 path: dwconv(squeeze=True) already handed back the mean
            count = 1
        p.add("se_gate", [sums], gate, consts, cite=cite, C=c, rd=rd, inv_count=1.0 / count, act=act,
              gate_act=gate_act, sums_fixed=1 if fixed else 0)
        return gate

    def scale_channels(self, x: TRef, gate: TRef, residual: Optional[TRef] = None, relu_after=False, cite="") -> TRef:
        out = self.p.new_tensor(x.rows, x.C, x.H, x.W, name="se_scaled")
        ins = [x, gate] + ([residual] if residual is not None else [])
        self.p.add("scale_channels", ins, out, cite=cite, R=x.rows, C=x.C, has_residual=residual is not None,
                   act_after=1 if relu_after else 0)
        return out

    def group_norm(self, x: TRef, prefix: str, groups: int, eps: float, *, act="", residual: Optional[TRef] = None,
                   act_after="", cite="") -> TRef:
        """GroupNormalization (layers/norm.py:37-165) + activation (+ shortcut add + activation)."""
        p = self.p
        g, b = self.wget(prefix + "/gamma"), self.wget(prefix + "/beta")
        assert g.shape[0] == x.C and x.C % groups == 0, f"{prefix}: {x.C} channels, {groups} groups"
        out = p.new_tensor(x.rows, x.C, x.H, x.W, name=prefix)
        ws = p.new_tensor(1, 4 * groups, dtype="f32", name=prefix + ":stats")       # int64 [groups][2] fixed-point sums
        consts = {"gamma": p.new_const(g, prefix + "/gamma"), "beta": p.new_const(b, prefix + "/beta")}
        ins = [x] + ([residual] if residual is not None else [])
        if residual is not None:
            assert residual.rows == x.rows and residual.C == x.C
        p.add("group_norm", ins, out, consts, cite=cite, extra_outputs=[ws], rows=x.rows, C=x.C, groups=groups,
              eps=float(eps), act=act, act_after=act_after, has_residual=residual is not None, ws=ws.id)
        return out

    def blur_pool(self, x: TRef, stride: int, cite="") -> TRef:
        """BlurPool2D(kernel_size=3, stride) (layers/blurpool.py:5-66)."""
        pad = (3 + stride) // 2 - 1
        OH, OW = (x.H + 2 * pad - 3) // stride + 1, (x.W + 2 * pad - 3) // stride + 1
        out = self.p.new_tensor(OH * OW, x.C, OH, OW, name="blur_pool")
        self.p.add("blur_pool", [x], out, cite=cite, H=x.H, W=x.W, C=x.C, stride=stride)
        return out

    def avg_pool(self, x: TRef, k: int, stride: int, cite="") -> TRef:
        """AveragePooling2D(k, stride, "same"): border windows average their valid elements (resnet.py:299-301)."""
        OH, OW = -(-x.H // stride), -(-x.W // stride)
        out = self.p.new_tensor(OH * OW, x.C, OH, OW, name="avg_pool")
        self.p.add("avg_pool", [x], out, cite=cite, H=x.H, W=x.W, C=x.C, k=k, stride=stride)
        return out

    def eca_gate(self, mean: TRef, kernel: str, gate_act="sigmoid", cite="") -> TRef:
        """EcaModule gate from fp32 channel means (layers/attention.py:110-126)."""
        assert mean.dtype == "f32" and mean.rows == 1
        w = np.ascontiguousarray(self.wget(kernel).reshape(-1), dtype=np.float32)      # Conv1D kernel (k, 1, 1)
        gate = self.p.new_tensor(1, mean.C, dtype="f32", name="eca_gate")
        self.p.add("eca_gate", [mean], gate, {"w": self.p.new_const(w, kernel)}, cite=cite, C=mean.C, k=int(w.shape[0]),
                   gate_act=gate_act)
        return gate

    def patch_merge_ln(self, x: TRef, prefix: str, eps: float, cite="") -> TRef:
        p = self.p
        g, b = self.wget(prefix + "/gamma"), self.wget(prefix + "/beta")
        assert g.shape[0] == 4 * x.C
        out = p.new_tensor((x.H // 2) * (x.W // 2), 4 * x.C, x.H // 2, x.W // 2, name="patch_merge")
        consts = {"gamma": p.new_const(g, prefix + "/gamma"), "beta": p.new_const(b, prefix + "/beta")}
        p.add("patch_merge_ln", [x], out, consts, cite=cite, H=x.H, W=x.W, C=x.C, eps=float(eps))
        return out

    def reshape(self, x: TRef, rows: int, C: int, H=0, W=0) -> TRef:
        """Free reinterpretation of a contiguous tensor (tf.reshape)."""
        assert rows * C == x.rows * x.C
        out = TRef(x.id, rows, C, H, W, x.dtype, x.keep, x.name)
        return out


# ---------------------------------------------------------------------------------------
# Plan: a program bound to device buffers for one batch size
# ---------------------------------------------------------------------------------------
class Plan:
    def __init__(self, prog: Program, batch: int, device: str = "cuda", external: Optional[Dict[int, int]] = None):
        """``device="cpu"`` builds the same call list over host buffers; it can only be used
        to inspect / marshal-check the plan (tests without a GPU) -- launching it fails.
        ``external``: tensor id -> device address: those tensors live at that address instead of in this plan's slabs (the
        branches of ``CapturedHybrid`` write what crosses the join straight into the full-batch plan's buffers)."""
        import torch
        from . import ffi
        self.ffi = ffi
        self.external = dict(external or {})
        self.op_call_start: List[int] = []      # first entry of ``calls`` of every op (CapturedHybrid cuts between ops)
        self.prog = prog
        self.batch = batch
        self.device = device
        prog.upload(device)
        assign, slabs = prog.plan_buffers()
        self.slab_bytes = [s * batch for s in slabs]
        self.slabs = [torch.empty(max(n, 16), dtype=torch.uint8, device=device) for n in self.slab_bytes]
        self.assign = assign
        self._keepalive = []
        self.calls: List[Tuple[Callable, tuple]] = []
        self._input_patch = None
        self._input_call = None
        self._stem_raw = None
        self._gemm_descs = []
        self._gemm_call_index = []
        self._tune_times = {}          # shape key -> {hint: ms} of the last isolated autotune
        self._build()
        if prog.precision == "fp32":
            self._bind_fp32()
        elif device != "cpu" and tune.autotune_enabled():
            self.autotune()

    # pointers ----------------------------------------------------------------------------------
    def tptr(self, tid: int) -> int:
        if tid in self.external:
            return self.external[tid]
        return self.slabs[self.assign[tid]].data_ptr()

    def cptr(self, cid: Optional[int]) -> Optional[int]:
        if cid is None:
            return None
        return self.prog._dev_consts[cid].data_ptr()

    def tensor_view(self, t: TRef):
        """torch view of an activation tensor (for outputs / features)."""
        import torch
        slab = self.slabs[self.assign[t.id]]
        n = self.batch * t.rows * t.C
        if t.dtype == "f32":
            return slab[: n * 4].view(torch.float32).view(self.batch, t.rows, t.C)
        return slab[: n * 2].view(torch.bfloat16).view(self.batch, t.rows, t.C)

    # build ---------------------------------------------------------------------------------------
    def _build(self):
        ffi, lib, B = self.ffi, self.ffi.lib, self.batch
        prog = self.prog
        for op in prog.ops:
            a = op.attrs
            k = op.kind
            self.op_call_start.append(len(self.calls))
            if k == "cast_input":
                out = self.tptr(op.output)
                H, W, cin = prog.input_shape
                self._input_patch = (len(self.calls), out, B * H * W, a["c_in"], a["c_out"])
                if a["pad"] != (0, 0, 0, 0):
                    pt_, pb_, pl_, pr_ = a["pad"]
                    self._input_call = (lib.tfimm_hip_cast_input_pad, (out, B, H, W, a["c_in"], pt_, pb_, pl_, pr_))
                    self._input_call_u8 = (lib.tfimm_hip_preprocess_input_pad, self._input_call[1])
                else:
                    self._input_call = (lib.tfimm_hip_cast_input, (out, B * H * W, a["c_in"], a["c_out"]))
                    self._input_call_u8 = (lib.tfimm_hip_preprocess_input, self._input_call[1])
                self.calls.append((self._input_call[0], None))  # input pointer / dtype patched per call
            elif k == "gemm":
                d = ffi.GemmDesc()
                a_ptr = self.tptr(op.inputs[0])
                d.mode = a["mode"]
                d.M = a["M"] * B
                d.N, d.K = a["N"], a["K"]
                d.ldw = a["ldw"]
                if a["mode"] == 0:
                    d.lda = a["lda"]
                    a_ptr += a.get("a_byte_offset", 0)
                    d.rows_per_image = a.get("a_rows_per_image", 0)
                else:
                    d.B, d.H, d.W, d.Cin = B, a["H"], a["W"], a["Cin"]
                    d.KH, d.KW, d.stride = a["KH"], a["KW"], a["stride"]
                    d.pad_t, d.pad_l, d.OH, d.OW = a["pad_t"], a["pad_l"], a["OH"], a["OW"]
                    d.stride_w = a.get("stride_w", 0)
                    d.pix_pitch = a.get("pix_pitch", 0)
                    a_ptr += a.get("a_byte_offset", 0)
                d.a = a_ptr
                d.wt = self.cptr(op.consts["wt"])
                d.bias = self.cptr(op.consts.get("bias"))
                out_t = prog.tensors[op.output]
                out_ptr = self.tptr(out_t.id) + a.get("out_col", 0) * out_t.itemsize + a.get("out_byte_offset", 0)
                d.out = out_ptr
                d.ldc = a.get("ldc", out_t.C)
                d.out_f32 = a["out_f32"]
                d.act = ffi.ACT[a["act"]]
                d.act_after_res = 1 if a["act_after_res"] else 0
                idx = 1
                if a.get("ln"):
                    d.ln_stats = self.tptr(op.inputs[idx])
                    d.ln_c1 = self.cptr(op.consts["ln_c1"])
                    idx += 1
                if a.get("has_residual"):
                    d.residual = self.tptr(op.inputs[idx]) + a.get("res_byte_offset", 0)
                    idx += 1
                    d.ldr = a["ldr"]
                elif "residual" in op.consts:
                    d.residual = self.cptr(op.consts["residual"])
                    d.ldr = a["N"]
                d.res_mod = a.get("res_mod", 0)
                if a.get("has_scale"):
                    d.a_scale = self.tptr(op.inputs[idx])
                    d.rows_per_image = a["a_rows_per_image"]
                if a.get("remap"):
                    d.remap_in, d.remap_out, d.remap_off = a["remap"]
                if a.get("dual"):
                    du = a["dual"]
                    d.a2 = self.tptr(op.inputs[idx])
                    idx += 1
                    d.K2, d.lda2, d.a2_stride = du["K2"], du["lda2"], du["stride"]
                    d.a2_H, d.a2_W, d.a2_OH, d.a2_OW = du["H"], du["W"], du["OH"], du["OW"]
                    d.a2_window = du.get("window", 0)
                d.tile_hint = tune.lookup(d)
                self._keepalive.append(d)
                self._gemm_descs.append(d)
                self._gemm_call_index.append(len(self.calls))
                self.calls.append((lib.tfimm_hip_gemm, (C.byref(d),)))
            elif k == "conv_chain":
                d = ffi.ChainDesc()
                d.x = self.tptr(op.inputs[0])
                d.w1, d.b1 = self.cptr(op.consts["w1"]), self.cptr(op.consts["b1"])
                d.w2, d.b2 = self.cptr(op.consts["w2"]), self.cptr(op.consts["b2"])
                d.residual = self.tptr(op.inputs[1]) if a["has_residual"] else None
                if a.get("has_ds"):
                    d.ds_x, d.ds_w, d.ds_cin = self.tptr(op.inputs[-1]), self.cptr(op.consts["ds_w"]), 64
                d.out = self.tptr(op.output)
                d.B, d.H, d.W, d.Cin, d.KH, d.KW = B, a["H"], a["W"], a["Cin"], a["KH"], a["KW"]
                d.stride, d.pad_t, d.pad_l, d.OH, d.OW = a["stride"], a["pad"], a["pad"], a["OH"], a["OW"]
                d.C1, d.N2, d.ldw1, d.ldw2, d.ldr, d.ldc = a["C1"], a["N2"], a["ldw1"], a["ldw2"], a["N2"], a["N2"]
                d.act1, d.act2 = ffi.ACT[a["act1"]], ffi.ACT[a["act2"]]
                self._keepalive.append(d)
                self.calls.append((lib.tfimm_hip_conv_chain, (C.byref(d),)))
            elif k == "mlp_fused":
                d = ffi.MlpDesc()
                d.x, d.residual, d.out = self.tptr(op.inputs[0]), self.tptr(op.inputs[1]), self.tptr(op.output)
                d.w1, d.b1 = self.cptr(op.consts["w1"]), self.cptr(op.consts["b1"])
                d.w2, d.b2 = self.cptr(op.consts["w2"]), self.cptr(op.consts["b2"])
                d.M, d.C, d.hidden, d.act, d.eps = B * a["rows"], a["C"], a["hidden"], ffi.ACT[a["act"]], a["eps"]
                self._keepalive.append(d)
                self.calls.append((lib.tfimm_hip_mlp_fused, (C.byref(d),)))
            elif k == "expand_dwconv":
                d = ffi.ExpandDwDesc()
                d.x, d.y = self.tptr(op.inputs[0]), self.tptr(op.output)
                d.w1, d.b1 = self.cptr(op.consts["w1"]), self.cptr(op.consts["b1"])
                d.wdw, d.b2 = self.cptr(op.consts["wdw"]), self.cptr(op.consts["b2"])
                d.sum_out = None
                if a["sums"] is not None:
                    d.sum_out = self.tptr(a["sums"])
                    self.calls.append(("memset", (d.sum_out, B * a["C"] * 8)))
                d.B, d.H, d.W, d.Cin, d.C, d.Cpad = B, a["H"], a["W"], a["Cin"], a["C"], a["Cpad"]
                d.k, d.stride, d.pad_t, d.pad_l, d.OH, d.OW = a["k"], a["stride"], a["pad_t"], a["pad_l"], a["OH"], a["OW"]
                d.act1 = d.act2 = ffi.ACT[a["act"]]
                d.stem, d.img_h, d.img_w = a.get("stem", 0), a.get("img_h", 0), a.get("img_w", 0)
                self._keepalive.append(d)
                self.calls.append((lib.tfimm_hip_expand_dwconv, (C.byref(d),)))
            elif k == "grouped_conv":
                self.calls.append((lib.tfimm_hip_grouped_conv3x3,
                                   (self.tptr(op.inputs[0]), self.cptr(op.consts["w"]), self.cptr(op.consts["bias"]),
                                    self.tptr(op.output), B, a["H"], a["W"], a["C"], a["stride"], ffi.ACT[a["act"]])))
            elif k == "row_stats":
                self.calls.append((lib.tfimm_hip_row_stats,
                                   (self.tptr(op.inputs[0]), self.tptr(op.output), B * a["rows"], a["d"], a["d"], a["eps"])))
            elif k == "layernorm":
                xp = self.tptr(op.inputs[0]) + a["x_byte_offset"]
                self.calls.append((lib.tfimm_hip_layernorm,
                                   (xp, self.tptr(op.output) + a["y_byte_offset"], self.cptr(op.consts["gamma"]),
                                    self.cptr(op.consts["beta"]), B * a["rows"], a["d"], a["x_stride"],
                                    a["y_stride"], a["eps"])))
            elif k == "attention":
                d = ffi.AttnDesc()
                d.qkv = self.tptr(op.inputs[0])
                d.out = self.tptr(op.output)
                d.rel_bias = self.cptr(op.consts.get("rel_bias"))
                d.bias_log2 = self.cptr(op.consts.get("bias_log2"))
                d.batch, d.n_tokens, d.heads, d.hd = B, a["n_tokens"], a["heads"], a["hd"]
                d.scale = a["scale"]
                d.window, d.shift, d.res_h, d.res_w = a["window"], a["shift"], a["res_h"], a["res_w"]
                self._keepalive.append(d)
                self.calls.append((lib.tfimm_hip_attention, (C.byref(d),)))
            elif k == "talking_heads_attention":
                d = ffi.ThaDesc()
                d.qkv = self.tptr(op.inputs[0])
                d.out = self.tptr(op.output)
                # the head-mixing layers go by HOST pointer (copied into the kernel arguments at launch)
                host = {r: prog.consts[op.consts[r]].host for r in ("wl", "bl", "ww", "bw")}
                self._keepalive.append(host)
                d.proj_l_w, d.proj_l_b = host["wl"].ctypes.data, host["bl"].ctypes.data
                d.proj_w_w, d.proj_w_b = host["ww"].ctypes.data, host["bw"].ctypes.data
                d.proj_dev = self.cptr(op.consts["dev"])
                d.batch, d.n_tokens, d.heads, d.hd, d.scale = B, a["n_tokens"], a["heads"], a["hd"], a["scale"]
                self._keepalive.append(d)
                self.calls.append((lib.tfimm_hip_talking_heads_attention, (C.byref(d),)))
            elif k == "class_attention":
                dm = a["heads"] * a["hd"]
                self.calls.append((lib.tfimm_hip_class_attention,
                                   (self.tptr(op.inputs[0]), self.tptr(op.inputs[1]), self.tptr(op.output), B,
                                    a["n_tokens"], a["heads"], a["hd"], dm, 2 * dm, dm)))
            elif k == "copy_rows":
                self.calls.append((lib.tfimm_hip_copy_rows,
                                   (self.tptr(op.inputs[0]), self.tptr(op.output), B, a["src_rows"], a["dst_rows"],
                                    a["dst_row0"], a["d"])))
            elif k == "stem_pool":
                d = ffi.StemDesc()
                d.x, d.out = self.tptr(op.inputs[0]), self.tptr(op.output)
                d.wt, d.bias = self.cptr(op.consts["wt"]), self.cptr(op.consts["bias"])
                d.batch, d.Hp, d.Wp2, d.OH, d.OW, d.ldw = B, a["Hp"], a["Wp2"], a["OH"], a["OW"], a["ldw"]
                # the stem can read the caller's RGB image itself (border, 4th channel and rounding applied while its
                # LDS ring is filled): launch_input() then skips the conversion pass for float / bf16 inputs
                cast = next((o.attrs for o in prog.ops if o.kind == "cast_input" and o.output == op.inputs[0]), None)
                if cast is not None and cast["c_in"] == 3 and self.prog.input_shape[2] == 3:
                    ih, iw, _ = self.prog.input_shape
                    self._stem_raw = (d, d.x, ih, iw, cast["pad"][0], cast["pad"][2])
                self._keepalive.append(d)
                self.calls.append((lib.tfimm_hip_stem_conv_pool, (C.byref(d),)))
            elif k == "maxpool":
                self.calls.append((lib.tfimm_hip_maxpool,
                                   (self.tptr(op.inputs[0]), self.tptr(op.output), B, a["H"], a["W"], a["C"],
                                    a["k"], a["stride"], a["pad"], a["OH"], a["OW"])))
            elif k == "mean_rows":
                self.calls.append((lib.tfimm_hip_mean_rows,
                                   (self.tptr(op.inputs[0]), self.tptr(op.output), B, a["R"], a["C"], a["out_f32"])))
            elif k == "bcast_rows":
                self.calls.append((lib.tfimm_hip_bcast_rows,
                                   (self.cptr(op.consts["src"]), self.tptr(op.output), B, a["n_rows"], a["d"],
                                    a["dst_rows"])))
            elif k == "dwconv":
                sums_ptr = None
                if a["sums"] is not None:
                    sums_ptr = self.tptr(a["sums"])
                    self.calls.append(("memset", (sums_ptr, B * a["C"] * 8)))
                self.calls.append((lib.tfimm_hip_dwconv,
                                   (self.tptr(op.inputs[0]), self.cptr(op.consts["w"]), self.cptr(op.consts.get("bias")),
                                    self.tptr(op.output), sums_ptr, B, a["H"], a["W"], a["C"], a["k"], a["stride"],
                                    a["pad_t"], a["pad_l"], a["OH"], a["OW"], ffi.ACT[a["act"]])))
            elif k == "se_gate":
                self.calls.append((lib.tfimm_hip_se_gate,
                                   (self.tptr(op.inputs[0]), a["sums_fixed"], a["inv_count"], self.cptr(op.consts["w1"]),
                                    self.cptr(op.consts["b1"]), self.cptr(op.consts["w2"]), self.cptr(op.consts["b2"]),
                                    self.tptr(op.output), B, a["C"], a["rd"], ffi.ACT[a["act"]], ffi.ACT[a["gate_act"]])))
            elif k == "scale_channels":
                res = self.tptr(op.inputs[2]) if a["has_residual"] else None
                self.calls.append((lib.tfimm_hip_scale_channels,
                                   (self.tptr(op.inputs[0]), self.tptr(op.inputs[1]), res, self.tptr(op.output), B,
                                    a["R"], a["C"], a["act_after"])))
            elif k == "attention_probs":
                self.calls.append((lib.tfimm_hip_attention_probs,
                                   (self.tptr(op.inputs[0]), self.tptr(op.output), B, a["n_tokens"], a["heads"], a["hd"],
                                    a["scale"])))
            elif k == "group_norm":
                res = self.tptr(op.inputs[1]) if a["has_residual"] else None
                self.calls.append((lib.tfimm_hip_group_norm,
                                   (self.tptr(op.inputs[0]), self.cptr(op.consts["gamma"]), self.cptr(op.consts["beta"]),
                                    res, self.tptr(op.output), self.tptr(a["ws"]), B, a["rows"], a["C"], a["groups"],
                                    a["eps"], ffi.ACT[a["act"]], ffi.ACT[a["act_after"]])))
            elif k == "blur_pool":
                self.calls.append((lib.tfimm_hip_blur_pool,
                                   (self.tptr(op.inputs[0]), self.tptr(op.output), B, a["H"], a["W"], a["C"], a["stride"])))
            elif k == "avg_pool":
                self.calls.append((lib.tfimm_hip_avg_pool,
                                   (self.tptr(op.inputs[0]), self.tptr(op.output), B, a["H"], a["W"], a["C"], a["k"],
                                    a["stride"])))
            elif k == "eca_gate":
                self.calls.append((lib.tfimm_hip_eca_gate,
                                   (self.tptr(op.inputs[0]), 1.0, self.cptr(op.consts["w"]), self.tptr(op.output), B,
                                    a["C"], a["k"], ffi.ACT[a["gate_act"]])))
            elif k == "patch_merge_ln":
                self.calls.append((lib.tfimm_hip_patch_merge_ln,
                                   (self.tptr(op.inputs[0]), self.tptr(op.output), self.cptr(op.consts["gamma"]),
                                    self.cptr(op.consts["beta"]), B, a["H"], a["W"], a["C"], a["eps"])))
            else:
                raise NotImplementedError(k)

    # kernels that are float32 on both paths
    _FP32_SHARED = ("tfimm_hip_se_gate", "tfimm_hip_eca_gate")

    def _bind_fp32(self):
        """Verification path (engine/precision.py): every call of the list goes to the float32 kernel of the same name
        (csrc/ref32.hip: tfimm_hip_ref_*, same arguments); ops that only exist as bf16 fusions must not be in the program."""
        lib = self.ffi.lib
        calls = []
        for fn, args in self.calls:
            if fn == "memset":
                calls.append((fn, args))
                continue
            name = fn.__name__
            if name in self._FP32_SHARED:
                calls.append((fn, args))
                continue
            ref = getattr(lib, name.replace("tfimm_hip_", "tfimm_hip_ref_", 1), None)
            if ref is None or not name.startswith("tfimm_hip_"):
                raise NotImplementedError(f"{name} has no float32 counterpart: the fp32 lowering must not emit it")
            calls.append((ref, args))
        self.calls = calls
        out, n_pixels, c_in, c_out = self._input_patch[1:5]
        self._input_call = (lib.tfimm_hip_ref_cast_input, (out, n_pixels, c_in, c_out))
        self._input_call_u8 = None
        self._stem_raw = None

    def autotune(self, iters: int = 3, verbose: bool = False) -> int:
        """Time every GEMM launch of this plan with each tile shape of the LDS-DMA families (on the
        plan's own buffers) and keep the fastest (tfimm_gemm_desc.tile_hint).  Results are cached
        per problem shape in ``tune`` for later plans.  Returns the number of shapes tuned."""
        import torch
        lib = self.ffi.lib
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        tuned = 0
        for d in self._gemm_descs:
            key = tune.key_of(d)
            if key in tune.TABLE:
                d.tile_hint = tune.TABLE[key]
                continue
            best, best_ms = 0, float("inf")
            times = self._tune_times.setdefault(key, {})
            for hint in tune.candidates_for(d):
                d.tile_hint = hint
                if lib.tfimm_hip_gemm(C.byref(d), st) != 0:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    lib.tfimm_hip_gemm(C.byref(d), st)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / iters
                if verbose:
                    print(f"tune {key} hint={hint} {ms * 1e3:.1f} us")
                times[hint] = ms
                if ms < best_ms:
                    best, best_ms = hint, ms
            tune.TABLE[key] = best
            d.tile_hint = best
            tuned += 1
        return tuned

    def autotune_in_context(self, x_dev, top: int = 4, iters: int = 3, verbose: bool = False,
                            challengers: Optional[Tuple[int, ...]] = None, min_gain: float = 0.0) -> int:
        """Second tuning pass.  ``autotune`` times a GEMM back to back with itself -- operands warm in
        L2 / Infinity Cache, which flatters tiles that re-read them.  Here the ``top`` fastest tiles of
        that pass are timed INSIDE a full forward (HIP events around the one launch, every other layer
        running as it will), summed over all layers that share the problem shape; the table keeps the
        winner.  Returns the number of shapes whose choice changed.

        ``challengers``: instead of the isolated pass' ranking, time the table's CURRENT choice against these hints (a new
        kernel entering an existing table: tools/retune_with.py); a challenger replaces the incumbent only when it is at
        least ``min_gain`` (fraction) faster."""
        import torch
        groups: Dict[str, List[int]] = {}
        for gi, d in enumerate(self._gemm_descs):
            groups.setdefault(tune.key_of(d), []).append(gi)
        stream_ptr = torch.cuda.current_stream().cuda_stream
        st = C.c_void_p(stream_ptr)
        idx = self._input_patch[0]

        def forward_timed(call_ids):
            evs = {}
            for i, (fn, args) in enumerate(self.calls):
                if i in call_ids:
                    evs[i] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    evs[i][0].record()
                if i == idx:
                    rc = self.launch_input(x_dev, st)
                elif fn == "memset":
                    _hip_memset_async(args[0], args[1], stream_ptr)
                    rc = 0
                else:
                    rc = fn(*args, st)
                if rc != 0:
                    return None
                if i in call_ids:
                    evs[i][1].record()
            torch.cuda.synchronize()
            return sum(a.elapsed_time(b) for a, b in evs.values())

        changed = 0
        for key, members in groups.items():
            if challengers is not None:
                allowed = tune.candidates_for(self._gemm_descs[members[0]])
                cands = [tune.TABLE.get(key, 0)] + [h for h in challengers if h in allowed and h != tune.TABLE.get(key, 0)]
                if len(cands) < 2:
                    continue
            else:
                times = self._tune_times.get(key)
                if not times or len(times) < 2:
                    continue
                cands = [h for h, _ in sorted(times.items(), key=lambda kv: kv[1])[:top]]
            call_ids = {self._gemm_call_index[gi] for gi in members}
            best, best_ms = tune.TABLE.get(key, 0), float("inf")
            for ci, hint in enumerate(cands):
                for gi in members:
                    self._gemm_descs[gi].tile_hint = hint
                ms = [forward_timed(call_ids) for _ in range(iters + 1)][1:]
                if any(m is None for m in ms):
                    continue
                m = min(ms)
                if verbose:
                    print(f"in-context {key} hint={hint} {m * 1e3:.1f} us")
                if m < best_ms * (1.0 - (min_gain if (challengers is not None and ci > 0) else 0.0)):
                    best, best_ms = hint, m
            if best != tune.TABLE.get(key):
                changed += 1
            tune.TABLE[key] = best
            for gi in members:
                self._gemm_descs[gi].tile_hint = best
        return changed

    def export(self) -> bytes:
        """This plan as a self-contained blob for the program-level C entry points (csrc/plan.hip, include/tfimm_hip.h:
        tfimm_hip_plan_query / _create / _forward / _output): every call with its arguments, the packed constants, the slab
        sizes and the named outputs.  Device pointers are written symbolically -- (slab, offset), (constant, offset), the
        caller's input -- and resolved against the caller's workspace by tfimm_hip_plan_create, so a host without Python runs
        exactly the launches this plan would (bit-identical results)."""
        import struct
        import torch
        if self.prog.precision != "bf16":
            raise NotImplementedError("plans of the float32 verification path are not exported")
        if self.device == "cpu":
            raise RuntimeError("export needs a plan built on the GPU (tile hints and occupancy are resolved there)")
        consts = self.prog._dev_consts
        hosts: List[np.ndarray] = []
        for k in self._keepalive:
            if isinstance(k, dict):
                hosts.extend(v for v in k.values() if isinstance(v, np.ndarray))
        ranges = ([(t.data_ptr(), max(t.numel() * t.element_size(), 1), 3, i) for i, t in enumerate(self.slabs)]
                  + [(t.data_ptr(), max(t.numel() * t.element_size(), 1), 4, i) for i, t in enumerate(consts)]
                  + [(h.ctypes.data, max(h.nbytes, 1), 5, i) for i, h in enumerate(hosts)])

        def ptr_ref(v):
            """(kind, aux, value) of a raw pointer value: NULL, or an offset into a slab / constant / host array"""
            if not v:
                return (2, 0, 0)
            for base, size, kind, idx in ranges:
                if base <= v < base + size:
                    return (kind, idx, v - base)
            for base, size, kind, idx in ranges:          # one past the end of a buffer (never dereferenced)
                if v == base + size:
                    return (kind, idx, v - base)
            raise ValueError(f"pointer {v:#x} is outside every buffer of the plan")

        structs: List[Tuple[bytes, list]] = []
        struct_ids: Dict[int, int] = {}

        def struct_ref(obj):
            if id(obj) in struct_ids:
                return struct_ids[id(obj)]
            raw = bytearray(bytes(obj))
            relocs = []
            for fname, ftype in obj._fields_:
                if ftype is C.c_void_p:
                    off = getattr(type(obj), fname).offset
                    relocs.append((off,) + ptr_ref(getattr(obj, fname)))
                    raw[off:off + 8] = b"\0" * 8
            structs.append((bytes(raw), relocs))
            struct_ids[id(obj)] = len(structs) - 1
            return len(structs) - 1

        def arg_ref(a, ctype):
            if ctype is C.c_void_p:
                return ptr_ref(a if isinstance(a, int) or a is None else getattr(a, "value", a))
            if hasattr(ctype, "_type_") and isinstance(ctype._type_, type) and issubclass(ctype._type_, C.Structure):
                return (7, struct_ref(a._obj), 0)
            if ctype is C.c_float:
                return (1, 0, struct.unpack("<Q", struct.pack("<d", float(a)))[0])
            return (0, 0, int(a) & 0xFFFFFFFFFFFFFFFF)

        stem_call = stem_desc = None
        pad_t = pad_l = 0
        if self._stem_raw is not None:
            # the fused ResNet stem can read the caller's image itself: exported pointing at the converted copy, the executor
            # re-points it per forward (tfimm_hip_plan_forward)
            stem_desc, padded_ptr, _, _, pad_t, pad_l = self._stem_raw
            stem_live = (stem_desc.x, stem_desc.in_dtype)          # the live plan keeps its own setting: restored below
            stem_desc.x, stem_desc.in_dtype = padded_ptr, 0
            stem_call = next(i for i, (fn, a) in enumerate(self.calls) if a and getattr(a[0], "_obj", None) is stem_desc)
        calls = []
        idx = self._input_patch[0]
        for i, (fn, args) in enumerate(self.calls):
            if fn == "memset":
                calls.append(("memset", [ptr_ref(args[0]), (0, 0, int(args[1]))]))
                continue
            if i == idx:
                fn = self._input_call[0]
                args = (None, 0) + tuple(self._input_call[1])
                refs = [(6, 0, 0)] + [arg_ref(a, t) for a, t in list(zip(args, fn.argtypes))[1:]]
            else:
                refs = [arg_ref(a, t) for a, t in zip(args, fn.argtypes)]
            calls.append((fn.__name__, refs))
        stem_struct = struct_ids[id(stem_desc)] if stem_desc is not None else -1
        stem_call = -1 if stem_call is None else stem_call
        if stem_desc is not None:
            stem_desc.x, stem_desc.in_dtype = stem_live            # (every struct has been serialised by now)
        H, W, cin = self.prog.input_shape

        def build(const_offsets, host_offsets):
            out = bytearray()
            out += struct.pack("<IIIIiii", 0x4c504654, 1, self.ffi.lib.tfimm_hip_abi_version(), self.batch, H, W, cin)
            out += struct.pack("<I", len(self.slabs)) + b"".join(struct.pack("<Q", n) for n in self.slab_bytes)
            out += struct.pack("<I", len(consts))
            for t, off in zip(consts, const_offsets):
                out += struct.pack("<QQ", t.numel() * t.element_size(), off)
            out += struct.pack("<I", len(hosts))
            for h, off in zip(hosts, host_offsets):
                out += struct.pack("<QQ", h.nbytes, off)
            out += struct.pack("<I", len(structs))
            for raw, relocs in structs:
                out += struct.pack("<II", len(raw), len(relocs)) + raw
                for off, kind, aux, value in relocs:
                    out += struct.pack("<IIIQ", off, kind, aux, value)
            out += struct.pack("<I", len(calls))
            for name, refs in calls:
                nb = name.encode()
                out += struct.pack("<I", len(nb)) + nb + struct.pack("<I", len(refs))
                for kind, aux, value in refs:
                    out += struct.pack("<IIQ", kind, aux, value)
            out += struct.pack("<iiiii", idx, stem_call, stem_struct, pad_t, pad_l)
            out += struct.pack("<I", len(self.prog.outputs))
            for name, t in self.prog.outputs.items():
                nb = name.encode()
                out += struct.pack("<I", len(nb)) + nb
                out += struct.pack("<IQQQI", self.assign[t.id], 0, t.rows, t.C, 1 if t.dtype == "f32" else 0)
            return out

        head = build([0] * len(consts), [0] * len(hosts))
        pos = (len(head) + 255) // 256 * 256
        const_offsets, host_offsets, data = [], [], []
        for t in consts:
            b = t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes() if t.numel() else b""
            const_offsets.append(pos)
            data.append((pos, b))
            pos = (pos + len(b) + 255) // 256 * 256
        for h in hosts:
            b = np.ascontiguousarray(h).tobytes()
            host_offsets.append(pos)
            data.append((pos, b))
            pos = (pos + len(b) + 255) // 256 * 256
        blob = bytearray(pos)
        head = build(const_offsets, host_offsets)
        blob[:len(head)] = head
        for off, b in data:
            blob[off:off + len(b)] = b
        return bytes(blob)

    def capture(self, x_dev, norm=None, sink=None) -> "CapturedPlan":
        return CapturedPlan(self, x_dev, norm, sink=sink)

    def check_marshalling(self):
        """Convert every recorded argument through the ctypes prototypes (no launch): catches
        arity / type slips in the call list without a GPU.  Returns the number of calls."""
        n = 0
        for i, (fn, args) in enumerate(self.calls):
            if fn == "memset":
                assert len(args) == 2
                continue
            if args is None and self.prog.precision == "fp32":
                args = (0, 0) + tuple(self._input_call[1]) + (None, None)
                fn = self._input_call[0]
            elif args is None:  # cast_input: patched per call
                args = (0, 0) + tuple(self._input_call[1])
                c_in = self._input_patch[3]
                u8_fn, u8_args = self._input_call_u8
                u8_args = (0,) + tuple(u8_args) + ((C.c_float * c_in)(), (C.c_float * c_in)())
                if len(u8_args) + 1 != len(u8_fn.argtypes):
                    raise TypeError(f"call {i} ({u8_fn.__name__}): {len(u8_args) + 1} args for {len(u8_fn.argtypes)}")
                for a, pt in zip(u8_args, u8_fn.argtypes):
                    pt.from_param(a)
            protos = fn.argtypes
            if len(args) + 1 != len(protos):
                raise TypeError(f"call {i} ({fn.__name__}): {len(args) + 1} args for {len(protos)} parameters")
            for a, pt in zip(args, protos):
                pt.from_param(a)
            n += 1
        return n

    # run -------------------------------------------------------------------------------------------
    def launch_input(self, x_dev, st, norm=None, force_convert=False) -> int:
        """The input step of the program: convert the caller's image into the engine's padded bf16 layout
        (tfimm_hip_cast_input[_pad], or tfimm_hip_preprocess_input[_pad] for uint8 + ``norm``) -- or, when the
        program starts with the fused ResNet stem and the image is RGB float / bf16, just point that kernel at the
        caller's image (``force_convert`` keeps the separate pass, e.g. to time it)."""
        import torch
        c_in = self._input_patch[3]
        if (x_dev.dtype == torch.uint8) != (norm is not None):
            raise TypeError("uint8 input needs norm=(mean, std); float input must not pass it")
        if self.prog.precision == "fp32":
            mean = std = None
            if norm is not None:
                mean = (C.c_float * c_in)(*[float(v) for v in norm[0]])
                std = (C.c_float * c_in)(*[float(v) for v in norm[1]])
            in_dtype = 2 if norm is not None else (1 if x_dev.dtype == torch.bfloat16 else 0)
            return self._input_call[0](x_dev.data_ptr(), in_dtype, *self._input_call[1], mean, std, st)
        if self._stem_raw is not None:
            d, padded_ptr, ih, iw, pad_t, pad_l = self._stem_raw
            raw = (norm is None and not force_convert and x_dev.dtype in (torch.bfloat16, torch.float32)
                   and os.environ.get("TFIMM_NO_STEM_RAW", "0") != "1")
            if raw:
                d.x = x_dev.data_ptr()
                d.in_dtype = 1 if x_dev.dtype == torch.bfloat16 else 2
                d.H, d.W, d.pad_t, d.pad_l = ih, iw, pad_t, pad_l
                return 0
            d.x, d.in_dtype = padded_ptr, 0
        if norm is not None:
            mean = (C.c_float * c_in)(*[float(v) for v in norm[0]])
            std = (C.c_float * c_in)(*[float(v) for v in norm[1]])
            return self._input_call_u8[0](x_dev.data_ptr(), *self._input_call_u8[1], mean, std, st)
        in_dtype = 1 if x_dev.dtype == torch.bfloat16 else 0
        return self._input_call[0](x_dev.data_ptr(), in_dtype, *self._input_call[1], st)

    def run(self, x_dev, stream_ptr: Optional[int] = None, norm=None, lo: int = 0, hi: Optional[int] = None):
        """Enqueue the whole program on the current torch stream.  ``x_dev``: contiguous cuda
        tensor (B, H, W, C) float32 or bfloat16 -- or uint8 with ``norm = (mean, std)`` (one float per
        channel): the model's preprocessing then runs inside the input conversion
        (tfimm_hip_preprocess_input).  ``lo`` / ``hi``: only entries [lo, hi) of ``calls`` (CapturedHybrid)."""
        import torch
        ffi = self.ffi
        if stream_ptr is None:
            stream_ptr = torch.cuda.current_stream().cuda_stream
        st = C.c_void_p(stream_ptr)
        idx = self._input_patch[0]
        # TFIMM_ROCTX=1: one roctx range per launch, named "<index> <op kind> <reference file:line>" -- rocprofv3
        # --marker-trace then attributes kernel time to the reference call site each launch replaces
        marks = self._roctx_labels() if os.environ.get("TFIMM_ROCTX", "0") == "1" else None
        hi = len(self.calls) if hi is None else hi
        for i in range(lo, hi):
            fn, args = self.calls[i]
            if marks is not None:
                ffi.roctx_push(marks[i])
            try:
                if i == idx:
                    rc = self.launch_input(x_dev, st, norm)
                elif fn == "memset":
                    ffi.check(_hip_memset_async(args[0], args[1], stream_ptr), "hipMemsetAsync")
                    continue
                else:
                    rc = fn(*args, st)
            finally:
                if marks is not None:
                    ffi.roctx_pop()
            if rc != 0:
                ffi.check(rc, f"op {i} ({getattr(fn, '__name__', fn)})")

    def _roctx_labels(self) -> List[str]:
        """One label per entry of ``self.calls``: the op it belongs to (memsets sit in front of their op)."""
        labels, ops = [], iter(self.prog.ops)
        for fn, _ in self.calls:
            if fn == "memset":
                labels.append("memset (squeeze sums)")
                continue
            op = next(ops)
            labels.append(f"{len(labels)} {op.kind} {op.cite}".strip())
        return labels


class CapturedPlan:
    """A plan recorded into a HIP graph (through torch.cuda.CUDAGraph, which drives
    hipStreamBeginCapture on the stream the C ABI launches on): one ``hipGraphLaunch`` per forward
    instead of ~100 ctypes calls + kernel launches.  The input pointer is baked into the graph, so
    callers either keep writing into ``static_input`` or replay on the tensor that was captured."""

    def __init__(self, plan: "Plan", x_dev, norm=None, sink=None):
        import torch
        self.plan = plan
        self.static_input = x_dev
        # every lazy host-side initialisation (function attributes, occupancy queries) must have
        # happened before capture: run the plan once eagerly
        plan.run(x_dev, norm=norm)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: only this thread's calls are recorded / policed -- a communicator's watchdog thread polling its
        # events (one process per GPU, RCCL initialised before the recording) must not invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            plan.run(x_dev, norm=norm)       # (mean/std travel by value in the recorded launch)
            _record_sink(sink, lambda t: plan.tensor_view(t))

    def replay(self):
        self.graph.replay()


class CapturedBranches:
    """Several plans, each over its own slice of ONE batch, recorded into one HIP graph as PARALLEL branches: the recording
    stream forks into a side stream per further plan and joins them at the end, so a replay is still one hipGraphLaunch.
    Kernels of different branches run side by side wherever a launch leaves compute units idle (196 tiles on 256 CUs, the
    tail of every launch) and an HBM-bound launch of one branch overlaps an MFMA-bound one of the other; every image is
    still computed by the same kernels on the same operands (results are bit-equal to the single plan's: the engine's
    kernels never mix images).  Measured at the scored batch sizes (tools/two_stream_probe.py): ResNet-50 +6 %, Swin-B +12 %,
    EfficientNet-B4 +-0, ViT-B/16 -4 % -- callers pick per workload (bench.py --branches auto times both)."""

    def __init__(self, plans: List["Plan"], x_dev, norm=None, sink=None):
        import torch
        self.plans = plans
        self.static_input = x_dev
        bounds = np.cumsum([0] + [p.batch for p in plans])
        assert bounds[-1] == x_dev.shape[0], (bounds, tuple(x_dev.shape))
        self.slices = [(int(bounds[i]), int(bounds[i + 1])) for i in range(len(plans))]
        for p, (lo, hi) in zip(plans, self.slices):      # lazy host-side initialisation happens outside the recording
            p.run(x_dev[lo:hi], norm=norm)
        torch.cuda.synchronize()
        self.side = [torch.cuda.Stream() for _ in plans[1:]]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            main = torch.cuda.current_stream()
            for s in self.side:
                s.wait_stream(main)                          # fork
            lo, hi = self.slices[0]
            plans[0].run(x_dev[lo:hi], norm=norm)
            for s, p, (lo, hi) in zip(self.side, plans[1:], self.slices[1:]):
                with torch.cuda.stream(s):
                    p.run(x_dev[lo:hi], norm=norm)
            for s in self.side:
                main.wait_stream(s)                          # join
            if sink is not None:                             # each branch's rows straight into the caller's tensor
                t, dst = sink
                for p, (lo, hi) in zip(plans, self.slices):
                    dst[lo:hi].copy_(p.tensor_view(t).view(dst[lo:hi].shape))

    def replay(self):
        self.graph.replay()

    def output(self, t: "TRef"):
        """The program output ``t`` of the whole batch (the branches' slices concatenated)."""
        import torch
        return torch.cat([p.tensor_view(t) for p in self.plans], dim=0)


class CapturedHybrid:
    """Parallel branches for the FIRST ops of the program only: two half-batch plans run ops [0, cut_op) side by side, the
    full-batch plan runs the rest.  Where a forward ends in launches too small to split (ResNet-50: the 14 x 14 and 7 x 7 stages
    lose 2 - 20 % as two half-batch launches, the 56 x 56 and 28 x 28 stages win 7 - 22 %: tools/stage_fork_probe.py) this keeps
    the gain of the early stages without the loss of the late ones.  What crosses the join lives in the FULL plan's buffers: the
    branches are built with those tensors ``external`` (their slice of the full tensor), so the join is free.  One
    ``hipGraphLaunch``, bit-equal results (tests/test_gpu_branches.py)."""

    def __init__(self, prog: "Program", x_dev, cut_op: int, norm=None, sink=None):
        import torch
        B = x_dev.shape[0]
        assert 0 < cut_op <= len(prog.ops) and B >= 2
        self.full = Plan(prog, B)
        nb = branch_sizes(B, 2)
        live = prog.live_across(cut_op) if cut_op < len(prog.ops) else [t.id for t in prog.tensors if t.keep and t.dtype != "raw"]
        live = [t for t in live if prog.tensors[t].dtype != "raw"]
        self.halves, lo = [], 0
        for n_img in nb:
            ext = {t: self.full.tptr(t) + lo * prog.tensors[t].bytes_per_image for t in live}
            self.halves.append(Plan(prog, n_img, external=ext))
            lo += n_img
        self.static_input = x_dev
        self.cut_op = cut_op
        self.slices = [(0, nb[0]), (nb[0], B)]
        cut = self.full.op_call_start[cut_op] if cut_op < len(prog.ops) else len(self.full.calls)
        assert all(h.op_call_start == self.full.op_call_start for h in self.halves)
        self.cut_call = cut

        def forward():
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)
            (a0, a1), (b0, b1) = self.slices
            self.halves[0].run(x_dev[a0:a1], norm=norm, hi=cut)
            with torch.cuda.stream(self.side):
                self.halves[1].run(x_dev[b0:b1], norm=norm, hi=cut)
            main.wait_stream(self.side)
            if cut < len(self.full.calls):
                self.full.run(x_dev, norm=norm, lo=cut)

        self.side = torch.cuda.Stream()
        self.full.run(x_dev, norm=norm)          # lazy host-side initialisation of every launch, outside the recording
        forward()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            forward()
            _record_sink(sink, lambda t: self.full.tensor_view(t))

    def replay(self):
        self.graph.replay()

    def output(self, t: "TRef"):
        return self.full.tensor_view(t)


def _record_sink(sink, view):
    """``sink = (TRef, tensor)``: the last node of a recording copies that program output into the caller's tensor, so a
    replay delivers it with the same ``hipGraphLaunch`` -- no copy kernel between two replays (bench.py: the logits that the
    exchange step sends; round 4 issued ``logits.copy_`` after every replay, a launch the next replay had to wait for)."""
    if sink is not None:
        t, dst = sink
        dst.copy_(view(t).view(dst.shape))


def _hip_memset_async(ptr: int, nbytes: int, stream_ptr: int) -> int:
    from . import ffi
    return ffi.lib.tfimm_hip_memset_async(C.c_void_p(ptr), 0, nbytes, C.c_void_p(stream_ptr))
