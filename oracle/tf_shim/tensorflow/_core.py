"""Tensor, Variable, dtypes, name scopes and the ``tf.*`` function surface of the stand-in.

TEST INFRASTRUCTURE (see package docstring).  Arithmetic with TF-specific semantics (SAME
padding, bicubic resize, roll direction, ...) delegates to ``oracle.ops`` so those rules are
written down exactly once and pinned by tests/test_oracle_ops.py; everything else is a thin
veneer over torch-CPU.
"""
import contextlib
import re

import numpy as np
import torch

from oracle import ops as _ops

# --------------------------------------------------------------------------------------------
# dtypes
# --------------------------------------------------------------------------------------------


class DType:
    def __init__(self, name, torch_dtype):
        self.name = name
        self.torch = torch_dtype

    @property
    def is_floating(self):
        return self.torch.is_floating_point

    def __repr__(self):
        return f"tf.{self.name}"

    def __eq__(self, other):
        try:
            return as_dtype(other).name == self.name
        except (TypeError, ValueError):
            return False

    def __hash__(self):
        return hash(self.name)


float16 = DType("float16", torch.float16)
bfloat16 = DType("bfloat16", torch.bfloat16)
float32 = DType("float32", torch.float32)
float64 = DType("float64", torch.float64)
int32 = DType("int32", torch.int32)
int64 = DType("int64", torch.int64)
uint8 = DType("uint8", torch.uint8)
bool_ = DType("bool", torch.bool)
_DTYPES = {d.name: d for d in (float16, bfloat16, float32, float64, int32, int64, uint8, bool_)}
_FROM_TORCH = {d.torch: d for d in _DTYPES.values()}


def as_dtype(d):
    if isinstance(d, DType):
        return d
    if isinstance(d, str):
        return _DTYPES[d]
    if isinstance(d, torch.dtype):
        return _FROM_TORCH[d]
    if d is float:
        return float32
    if d is int:
        return int32
    if d is bool:
        return bool_
    return _DTYPES[np.dtype(d).name]


# --------------------------------------------------------------------------------------------
# shapes
# --------------------------------------------------------------------------------------------


class TensorShape(tuple):
    """Static shape: a tuple of ints with the accessors the reference uses
    (``.as_list()``, ``.ndims`` / ``.rank``, slicing)."""

    def as_list(self):
        return list(self)

    @property
    def ndims(self):
        return len(self)

    rank = ndims

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return TensorShape(r) if isinstance(i, slice) else r


# --------------------------------------------------------------------------------------------
# tensors
# --------------------------------------------------------------------------------------------


def _raw(x, dtype=None):
    """Anything tensor-like -> torch tensor (python floats become fp32, ints int32 like TF)."""
    if isinstance(x, Tensor):
        t = x._t
    elif isinstance(x, torch.Tensor):
        t = x
    elif isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x))
    elif isinstance(x, (bool, np.bool_)):
        t = torch.tensor(bool(x))
    elif isinstance(x, (int, np.integer)):
        t = torch.tensor(int(x), dtype=torch.int32)
    elif isinstance(x, (float, np.floating)):
        t = torch.tensor(float(x), dtype=torch.float32)
    elif isinstance(x, (list, tuple)):
        if len(x) and any(isinstance(e, (Tensor, torch.Tensor)) for e in x):
            t = torch.stack([_raw(e) for e in x])
        else:
            a = np.asarray(x)
            if a.dtype == np.float64:
                a = a.astype(np.float32)
            elif a.dtype == np.int64:
                a = a.astype(np.int32)
            t = torch.from_numpy(np.ascontiguousarray(a))
    else:
        raise TypeError(f"cannot convert {type(x).__name__} to a tensor")
    if dtype is not None:
        t = t.to(as_dtype(dtype).torch)
    return t


def _binary(a, b):
    """TF binary-op operand rule: python scalars adopt the tensor operand's dtype."""
    ta = isinstance(a, (Tensor, torch.Tensor, np.ndarray))
    tb = isinstance(b, (Tensor, torch.Tensor, np.ndarray))
    if ta and not tb:
        ra = _raw(a)
        return ra, _raw(b).to(ra.dtype) if not isinstance(b, (list, tuple)) else _raw(b, _FROM_TORCH[ra.dtype])
    if tb and not ta:
        rb = _raw(b)
        return _raw(a).to(rb.dtype) if not isinstance(a, (list, tuple)) else _raw(a, _FROM_TORCH[rb.dtype]), rb
    return _raw(a), _raw(b)


class Tensor:
    __array_priority__ = 100

    def __init__(self, t):
        self._t = t

    # -- introspection -------------------------------------------------------------------------
    @property
    def shape(self):
        return TensorShape(int(d) for d in self._t.shape)

    def get_shape(self):
        return self.shape

    @property
    def dtype(self):
        return _FROM_TORCH[self._t.dtype]

    @property
    def ndim(self):
        return self._t.dim()

    def numpy(self):
        return self._t.detach().numpy().copy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __len__(self):
        return int(self._t.shape[0])

    def __iter__(self):
        for i in range(len(self)):
            yield Tensor(self._t[i])

    def __repr__(self):
        return f"<shim tf.Tensor shape={tuple(self._t.shape)} dtype={self.dtype.name}>"

    def __bool__(self):
        return bool(self._t)

    def __int__(self):
        return int(self._t)

    def __float__(self):
        return float(self._t)

    def __index__(self):
        return int(self._t)

    __hash__ = object.__hash__

    # -- indexing ------------------------------------------------------------------------------
    def __getitem__(self, idx):
        def conv(i):
            if isinstance(i, Tensor):
                return i._t if i._t.dim() else int(i._t)
            if isinstance(i, slice):
                return slice(*(None if v is None else int(v) for v in (i.start, i.stop, i.step)))
            return i
        idx = tuple(conv(i) for i in idx) if isinstance(idx, tuple) else conv(idx)
        return Tensor(self._t[idx])

    # -- arithmetic ----------------------------------------------------------------------------
    def _bin(self, other, fn, swap=False):
        a, b = _binary(other, self) if swap else _binary(self, other)
        return Tensor(fn(a, b))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, torch.true_divide)
    def __rtruediv__(self, o): return self._bin(o, torch.true_divide, True)
    def __floordiv__(self, o): return self._bin(o, torch.floor_divide)
    def __pow__(self, o): return self._bin(o, torch.pow)
    def __matmul__(self, o): return self._bin(o, torch.matmul)
    def __rmatmul__(self, o): return self._bin(o, torch.matmul, True)
    def __neg__(self): return Tensor(-self._t)
    def __eq__(self, o): return self._bin(o, torch.eq)
    def __ne__(self, o): return self._bin(o, torch.ne)
    def __lt__(self, o): return self._bin(o, torch.lt)
    def __le__(self, o): return self._bin(o, torch.le)
    def __gt__(self, o): return self._bin(o, torch.gt)
    def __ge__(self, o): return self._bin(o, torch.ge)


# --------------------------------------------------------------------------------------------
# name scopes and variables
# --------------------------------------------------------------------------------------------

_scope = []          # current name-scope stack (list of str)


@contextlib.contextmanager
def name_scope(name):
    _scope.append(name)
    try:
        yield
    finally:
        _scope.pop()


@contextlib.contextmanager
def fresh_name_scope():
    """Drop the enclosing scopes (what Keras' functional-construction graph does to layers
    that are built by a Sequential / functional model)."""
    saved = _scope[:]
    del _scope[:]
    try:
        yield
    finally:
        _scope[:] = saved


def current_scope():
    return "/".join(_scope)


class Variable(Tensor):
    """``tf.Variable``: eager variables are named ``<name scope>/<name>:0``."""

    def __init__(self, initial_value=None, trainable=True, name=None, dtype=None, shape=None, **_):
        if callable(initial_value):
            initial_value = initial_value()
        t = _raw(initial_value, dtype).clone()
        super().__init__(t)
        scope = current_scope()
        base = name or "Variable"
        self.name = (scope + "/" if scope else "") + base + ":0"
        self.trainable = trainable

    def assign(self, value):
        v = _raw(value).to(self._t.dtype)
        if tuple(v.shape) != tuple(self._t.shape):
            raise ValueError(f"Cannot assign value to variable '{self.name}': Shape mismatch. The variable shape "
                             f"{tuple(self._t.shape)}, and the assigned value shape {tuple(v.shape)} are incompatible.")
        self._t = v.clone()
        return self

    def value(self):
        return Tensor(self._t)

    def read_value(self):
        return Tensor(self._t)

    def __repr__(self):
        return f"<shim tf.Variable '{self.name}' shape={tuple(self._t.shape)} dtype={self.dtype.name}>"


# --------------------------------------------------------------------------------------------
# tf.* functions
# --------------------------------------------------------------------------------------------


def _ints(seq):
    return [int(v) for v in seq]


def convert_to_tensor(value, dtype=None, **_):
    if isinstance(value, np.ndarray):      # numpy keeps its own dtype (float64 stays float64)
        t = torch.from_numpy(np.ascontiguousarray(value))
        return Tensor(t if dtype is None else t.to(as_dtype(dtype).torch))
    return Tensor(_raw(value, dtype))


constant = convert_to_tensor


def cast(x, dtype):
    return Tensor(_raw(x).to(as_dtype(dtype).torch))


def shape(x):
    """Eager ``tf.shape``: plain python ints (the reference only ever unstacks it, slices it or
    feeds it back into ``tf.reshape``)."""
    if isinstance(x, (Tensor, torch.Tensor, np.ndarray)):
        return TensorShape(int(d) for d in _raw(x).shape)
    return TensorShape(np.asarray(x).shape)


def rank(x):
    return len(shape(x))


def reshape(tensor, shape, name=None):
    return Tensor(_raw(tensor).reshape(_ints(shape)))


def transpose(a, perm=None, **_):
    t = _raw(a)
    if perm is None:
        perm = list(reversed(range(t.dim())))
    return Tensor(t.permute(_ints(perm)).contiguous())


def unstack(value, num=None, axis=0):
    if isinstance(value, (tuple, list)) and not isinstance(value, Tensor):
        return list(value)
    return [Tensor(t) for t in torch.unbind(_raw(value), dim=axis)]


def stack(values, axis=0, **_):
    return Tensor(torch.stack([_raw(v) for v in values], dim=axis))


def concat(values, axis, **_):
    if all(isinstance(v, (tuple, list)) for v in values):     # shape arithmetic (layers/norm.py:87)
        out = []
        for v in values:
            out.extend(int(e) for e in v)
        return TensorShape(out)
    return Tensor(torch.cat([_raw(v) for v in values], dim=axis))


def expand_dims(input, axis, **_):
    t = _raw(input)
    if axis < 0:
        axis = t.dim() + 1 + axis
    return Tensor(t.unsqueeze(axis))


def squeeze(input, axis=None, **_):
    t = _raw(input)
    if axis is None:
        return Tensor(t.squeeze())
    for a in sorted([axis] if isinstance(axis, int) else list(axis), reverse=True):
        t = t.squeeze(a)
    return Tensor(t)


def zeros(shape, dtype=float32, **_):
    return Tensor(torch.zeros(_ints(shape), dtype=as_dtype(dtype).torch))


def ones(shape, dtype=float32, **_):
    return Tensor(torch.ones(_ints(shape), dtype=as_dtype(dtype).torch))


def zeros_like(x, dtype=None):
    t = _raw(x)
    return Tensor(torch.zeros_like(t, dtype=None if dtype is None else as_dtype(dtype).torch))


def ones_like(x, dtype=None):
    t = _raw(x)
    return Tensor(torch.ones_like(t, dtype=None if dtype is None else as_dtype(dtype).torch))


def range_(*args, dtype=None):
    t = torch.arange(*[int(a) for a in args], dtype=torch.int32)
    return Tensor(t if dtype is None else t.to(as_dtype(dtype).torch))


def repeat(input, repeats, axis=None):
    t = _raw(input)
    r = repeats if isinstance(repeats, int) else _raw(repeats)
    if isinstance(r, torch.Tensor) and r.dim() == 0:
        r = int(r)
    return Tensor(torch.repeat_interleave(t, r, dim=axis))


def tile(input, multiples):
    return Tensor(_raw(input).repeat(_ints(multiples)))


def split(value, num_or_size_splits, axis=0, **_):
    t = _raw(value)
    if isinstance(num_or_size_splits, int):
        assert t.shape[axis] % num_or_size_splits == 0
        parts = torch.split(t, t.shape[axis] // num_or_size_splits, dim=axis)
    else:
        parts = torch.split(t, _ints(num_or_size_splits), dim=axis)
    return [Tensor(p) for p in parts]


def roll(input, shift, axis):
    shift = [shift] if isinstance(shift, int) else _ints(shift)
    axis = [axis] if isinstance(axis, int) else _ints(axis)
    return Tensor(_ops.roll(_raw(input), shift, axis))


def gather(params, indices, axis=0, **_):
    p, i = _raw(params), _raw(indices).long()
    return Tensor(torch.index_select(p, axis, i.reshape(-1)).reshape(
        tuple(p.shape[:axis]) + tuple(i.shape) + tuple(p.shape[axis + 1:])))


def where(condition, x=None, y=None):
    c = _raw(condition).bool()
    if isinstance(x, (int, float)) and isinstance(y, (Tensor, torch.Tensor)):
        ry = _raw(y)
        return Tensor(torch.where(c, torch.tensor(x, dtype=ry.dtype), ry))
    if isinstance(y, (int, float)) and isinstance(x, (Tensor, torch.Tensor)):
        rx = _raw(x)
        return Tensor(torch.where(c, rx, torch.tensor(y, dtype=rx.dtype)))
    a, b = _binary(x, y)
    return Tensor(torch.where(c, a, b))


def pad(tensor, paddings, mode="CONSTANT", constant_values=0, **_):
    t = _raw(tensor)
    p = [[int(a), int(b)] for a, b in (np.asarray(paddings).tolist() if not isinstance(paddings, list) else paddings)]
    mode = mode.upper()
    if mode == "CONSTANT":
        flat = []
        for a, b in reversed(p):
            flat += [a, b]
        return Tensor(torch.nn.functional.pad(t, flat, value=constant_values))
    if mode == "REFLECT":       # mirror without repeating the edge element (layers/blurpool.py:53)
        for ax, (a, b) in enumerate(p):
            if a == 0 and b == 0:
                continue
            n = t.shape[ax]
            idx = list(range(a, 0, -1)) + list(range(n)) + list(range(n - 2, n - 2 - b, -1))
            t = torch.index_select(t, ax, torch.tensor(idx))
        return Tensor(t)
    raise NotImplementedError(mode)


def floor(x):
    return Tensor(torch.floor(_raw(x)))


def identity(x, **_):
    return Tensor(_raw(x))


def _axes(axis, nd):
    if axis is None:
        return tuple(range(nd))
    if isinstance(axis, int):
        return (axis,)
    return tuple(int(a) for a in axis)


def reduce_mean(input_tensor, axis=None, keepdims=False, **_):
    t = _raw(input_tensor)
    return Tensor(t.mean(dim=_axes(axis, t.dim()), keepdim=keepdims))


def reduce_sum(input_tensor, axis=None, keepdims=False, **_):
    t = _raw(input_tensor)
    return Tensor(t.sum(dim=_axes(axis, t.dim()), keepdim=keepdims))


def reduce_max(input_tensor, axis=None, keepdims=False, **_):
    t = _raw(input_tensor)
    return Tensor(torch.amax(t, dim=_axes(axis, t.dim()), keepdim=keepdims))


def reduce_variance(input_tensor, axis=None, keepdims=False, **_):
    t = _raw(input_tensor)
    ax = _axes(axis, t.dim())
    m = t.mean(dim=ax, keepdim=True)
    return Tensor(((t - m) ** 2).mean(dim=ax, keepdim=keepdims))


def sqrt(x):
    return Tensor(torch.sqrt(_raw(x)))


def rsqrt(x):
    return Tensor(torch.rsqrt(_raw(x)))


def divide(x, y):
    a, b = _binary(x, y)
    return Tensor(a / b)


def matmul(a, b, transpose_a=False, transpose_b=False, **_):
    ta, tb = _raw(a), _raw(b)
    if transpose_a:
        ta = ta.transpose(-1, -2)
    if transpose_b:
        tb = tb.transpose(-1, -2)
    return Tensor(ta @ tb)


def function(func=None, **_):
    """``tf.function``: eager execution only."""
    if func is None:
        return lambda f: f
    return func


class TensorSpec:
    def __init__(self, shape=None, dtype=float32, name=None):
        self.shape, self.dtype, self.name = shape, dtype, name


# --------------------------------------------------------------------------------------------
# helpers shared with the keras layer
# --------------------------------------------------------------------------------------------


def to_snake_case(name):
    """keras.utils.generic_utils.to_snake_case."""
    intermediate = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    insecure = re.sub("([a-z])([A-Z])", r"\1_\2", intermediate).lower()
    return insecure if insecure[0] != "_" else "private" + insecure
