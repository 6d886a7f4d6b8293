"""``tf.keras.layers`` of the stand-in: the Layer protocol (lazy build, name scopes, variable
tracking, ``training`` propagation) and the concrete layers the tfimm forward path instantiates.

TEST INFRASTRUCTURE.  Keras 2.12 behaviours restated here (third-party, not in /root/reference):
  * auto names: snake_case(class name) + ``_<n>`` per process-wide counter;
  * ``Layer.__call__`` enters ``name_scope(self.name)``, builds on first call with the input's
    static shape, then runs ``call``; variables created meanwhile are named
    ``<enclosing scopes>/<weight name>:0``;
  * a ``Sequential`` builds its layers in a fresh functional-construction graph, so their
    variables do NOT inherit the scopes that enclose the Sequential (this is why the reference
    spells full paths such as ``name + "/downsample/0"``, resnet.py:309-330);
  * ``training`` not passed explicitly is inherited from the enclosing layer call;
  * BatchNormalization(training=False) uses moving statistics; LayerNormalization normalises
    the last axis with population variance; Dense contracts the last axis.
"""
import inspect
from collections import OrderedDict

import numpy as np
import torch

from oracle import ops as _ops

from .. import _core as C
from .._core import Tensor, Variable
from . import activations as _act
from . import initializers as _init

_name_uids = {}
_call_ctx = []      # stack of (layer, training)


def reset_uids():
    _name_uids.clear()


def _unique_name(base):
    n = _name_uids.get(base, 0)
    _name_uids[base] = n + 1
    return base if n == 0 else f"{base}_{n}"


def _shape_of(x):
    if isinstance(x, (Tensor, torch.Tensor, np.ndarray)):
        return C.shape(x)
    if isinstance(x, (list, tuple)):
        return [_shape_of(e) for e in x]
    if isinstance(x, dict):
        return {k: _shape_of(v) for k, v in x.items()}
    return None


def _tensorise(x):
    if isinstance(x, Tensor):
        return x
    if isinstance(x, (np.ndarray, torch.Tensor)):
        t = C._raw(x)
        if t.dtype == torch.float64:        # Keras autocasts float inputs to the layer dtype (floatx)
            t = t.float()
        return Tensor(t)
    if isinstance(x, list):
        return [_tensorise(e) for e in x]
    if isinstance(x, tuple):
        return tuple(_tensorise(e) for e in x)
    return x


class Layer:
    def __init__(self, trainable=True, name=None, dtype=None, dynamic=False, **kwargs):
        kwargs.pop("input_shape", None)
        kwargs.pop("autocast", None)
        if kwargs:
            raise TypeError(f"{type(self).__name__}: unexpected keyword arguments {sorted(kwargs)}")
        object.__setattr__(self, "_own_vars", [])
        self.name = name if name else _unique_name(C.to_snake_case(type(self).__name__))
        self.trainable = trainable
        self.built = False
        self._dtype = dtype or "float32"
        params = list(inspect.signature(self.call).parameters.values())
        names = [p.name for p in params]
        self._call_has_training = "training" in names
        if self._call_has_training:
            self._training_pos = names.index("training")
            d = params[self._training_pos].default
            self._training_default = None if d is inspect.Parameter.empty else d

    # -- protocol ------------------------------------------------------------------------------
    @property
    def dtype(self):
        return self._dtype

    @property
    def compute_dtype(self):
        return self._dtype

    def build(self, input_shape):
        self.built = True

    def call(self, inputs, *args, **kwargs):
        return inputs

    def add_weight(self, name=None, shape=None, dtype=None, initializer=None, regularizer=None,
                   trainable=None, constraint=None, **kwargs):
        shape = tuple(int(d) for d in (shape if shape is not None else ()))
        init = _init.get(initializer if initializer is not None else "glorot_uniform")
        value = init(shape, dtype=dtype or "float32")
        v = Variable(value, trainable=True if trainable is None else trainable, name=name, dtype=dtype or "float32")
        self._own_vars.append(v)
        return v

    add_variable = add_weight

    def __call__(self, *args, **kwargs):
        if not args:
            raise ValueError("The first argument to `Layer.call` must always be passed.")
        args = (_tensorise(args[0]),) + args[1:]
        training = _call_ctx[-1][1] if _call_ctx else None
        if self._call_has_training:
            if kwargs.get("training") is not None:
                training = kwargs["training"]
            elif len(args) > self._training_pos:            # positional (ViT.call -> forward_features(x, training, ...))
                training = args[self._training_pos]
            else:
                kwargs.pop("training", None)
                if training is not None:                    # inherited from the enclosing layer call
                    kwargs["training"] = training
                else:
                    training = self._training_default
        elif "training" in kwargs:          # Keras drops the argument for layers whose call() does not take it
            t = kwargs.pop("training")
            training = t if t is not None else training
        with C.name_scope(self.name):
            if not self.built:
                self.build(_shape_of(args[0]))
                self.built = True
            _call_ctx.append((self, training))
            try:
                return self.call(*args, **kwargs)
            finally:
                _call_ctx.pop()

    # -- tracking ------------------------------------------------------------------------------
    def _children(self):
        """Layers and loose variables reachable from attributes, in attribute order."""
        seen, layers, loose = set(), [], []

        def visit(v):
            if isinstance(v, Layer):
                if id(v) not in seen:
                    seen.add(id(v))
                    layers.append(v)
            elif isinstance(v, Variable):
                if id(v) not in seen:
                    seen.add(id(v))
                    loose.append(v)
            elif isinstance(v, (list, tuple)):
                for e in v:
                    visit(e)
            elif isinstance(v, dict):
                for e in v.values():
                    visit(e)

        for k, v in list(self.__dict__.items()):
            if k == "_own_vars":
                continue
            visit(v)
        return layers, loose

    def _flatten_layers(self, include_self=True):
        out, seen = [], set()

        def rec(l):
            if id(l) in seen:
                return
            seen.add(id(l))
            out.append(l)
            for c in l._children()[0]:
                rec(c)
        rec(self)
        return out if include_self else out[1:]

    @property
    def layers(self):
        return self._children()[0]

    @property
    def weights(self):
        out, seen = [], set()
        for l in self._flatten_layers():
            for v in list(l._own_vars) + l._children()[1]:
                if id(v) not in seen:
                    seen.add(id(v))
                    out.append(v)
        return out

    variables = weights

    @property
    def trainable_weights(self):
        return [v for v in self.weights if v.trainable]

    trainable_variables = trainable_weights

    @property
    def non_trainable_weights(self):
        return [v for v in self.weights if not v.trainable]

    def get_weights(self):
        return [v.numpy() for v in self.weights]

    def set_weights(self, weights):
        ws = self.weights
        if len(ws) != len(weights):
            raise ValueError(f"You called `set_weights(weights)` on layer \"{self.name}\" with a weight list of "
                             f"length {len(weights)}, but the layer was expecting {len(ws)} weights.")
        for v, w in zip(ws, weights):
            v.assign(w)

    def count_params(self):
        return int(sum(int(np.prod(v.shape)) for v in self.weights))

    def get_config(self):
        return {"name": self.name}


# ---- concrete layers ------------------------------------------------------------------------------


class Activation(Layer):
    def __init__(self, activation, **kwargs):
        super().__init__(**kwargs)
        self.activation = _act.get(activation)

    def call(self, inputs):
        return self.activation(inputs)


class ReLU(Layer):
    def __init__(self, max_value=None, negative_slope=0.0, threshold=0.0, **kwargs):
        super().__init__(**kwargs)
        assert negative_slope == 0.0 and threshold == 0.0
        self.max_value = max_value

    def call(self, inputs):
        t = torch.relu(C._raw(inputs))
        if self.max_value is not None:
            t = torch.clamp(t, max=float(self.max_value))
        return Tensor(t)


class Dropout(Layer):
    def __init__(self, rate, noise_shape=None, seed=None, **kwargs):
        super().__init__(**kwargs)
        self.rate = rate

    def call(self, inputs, training=None):
        if training and self.rate > 0:
            raise NotImplementedError("the stand-in implements the inference path only")
        return inputs


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.units = int(units)
        self.activation = _act.get(activation)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel = None
        self.bias = None

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", shape=(input_shape[-1], self.units), initializer=self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", shape=(self.units,), initializer=self.bias_initializer)

    def call(self, inputs):
        y = _ops.dense(C._raw(inputs), self.kernel._t, self.bias._t if self.use_bias else None)
        return self.activation(Tensor(y))


def _tuple2(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", data_format=None,
                 dilation_rate=(1, 1), groups=1, activation=None, use_bias=True,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", kernel_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, **kwargs):
        super().__init__(**kwargs)
        assert data_format in (None, "channels_last")
        self.filters = int(filters)
        self.kernel_size = _tuple2(kernel_size)
        self.strides = _tuple2(strides)
        self.padding = padding.lower()
        self.dilation_rate = _tuple2(dilation_rate)
        self.groups = groups
        self.activation = _act.get(activation)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel = None
        self.bias = None

    def build(self, input_shape):
        cin = input_shape[-1]
        assert cin % self.groups == 0 and self.filters % self.groups == 0
        self.kernel = self.add_weight("kernel", shape=self.kernel_size + (cin // self.groups, self.filters),
                                      initializer=self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", shape=(self.filters,), initializer=self.bias_initializer)

    def call(self, inputs):
        y = _ops.conv2d(C._raw(inputs), C._raw(self.kernel), C._raw(self.bias) if self.use_bias else None,
                        stride=self.strides, padding=self.padding, groups=self.groups, dilation=self.dilation_rate)
        return self.activation(Tensor(y))


class DepthwiseConv2D(Layer):
    def __init__(self, kernel_size, strides=(1, 1), padding="valid", depth_multiplier=1, data_format=None,
                 dilation_rate=(1, 1), groups=1, activation=None, use_bias=True,
                 depthwise_initializer="glorot_uniform", bias_initializer="zeros", depthwise_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, depthwise_constraint=None,
                 bias_constraint=None, **kwargs):
        super().__init__(**kwargs)
        assert depth_multiplier == 1
        self.kernel_size = _tuple2(kernel_size)
        self.strides = _tuple2(strides)
        self.padding = padding.lower()
        self.dilation_rate = _tuple2(dilation_rate)
        self.activation = _act.get(activation)
        self.use_bias = use_bias
        self.depthwise_initializer = depthwise_initializer
        self.bias_initializer = bias_initializer
        self.depthwise_kernel = None
        self.bias = None

    def build(self, input_shape):
        c = input_shape[-1]
        self.depthwise_kernel = self.add_weight("depthwise_kernel", shape=self.kernel_size + (c, 1),
                                                initializer=self.depthwise_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", shape=(c,), initializer=self.bias_initializer)

    def call(self, inputs):
        y = _ops.depthwise_conv2d(C._raw(inputs), C._raw(self.depthwise_kernel),
                                  C._raw(self.bias) if self.use_bias else None, stride=self.strides,
                                  padding=self.padding, dilation=self.dilation_rate)
        return self.activation(Tensor(y))


class Conv1D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", data_format=None, dilation_rate=1,
                 groups=1, activation=None, use_bias=True, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        assert padding == "valid" and strides == 1 and dilation_rate == 1 and groups == 1
        self.filters = int(filters)
        self.kernel_size = int(kernel_size if isinstance(kernel_size, int) else kernel_size[0])
        self.activation = _act.get(activation)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel = None
        self.bias = None

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", shape=(self.kernel_size, input_shape[-1], self.filters),
                                      initializer=self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", shape=(self.filters,), initializer=self.bias_initializer)

    def call(self, inputs):
        x = C._raw(inputs)                                     # (N, L, Cin), cross-correlation over L
        w = C._raw(self.kernel).permute(2, 1, 0).contiguous()    # (k, Cin, Cout) -> (Cout, Cin, k)
        y = torch.nn.functional.conv1d(x.permute(0, 2, 1), w, C._raw(self.bias) if self.use_bias else None)
        return self.activation(Tensor(y.permute(0, 2, 1).contiguous()))


class ZeroPadding2D(Layer):
    def __init__(self, padding=(1, 1), data_format=None, **kwargs):
        super().__init__(**kwargs)
        if isinstance(padding, int):
            self.padding = ((padding, padding), (padding, padding))
        elif isinstance(padding[0], int):
            self.padding = ((padding[0], padding[0]), (padding[1], padding[1]))
        else:
            self.padding = (tuple(padding[0]), tuple(padding[1]))

    def call(self, inputs):
        return Tensor(_ops.zero_pad2d(C._raw(inputs), self.padding))


class ZeroPadding1D(Layer):
    def __init__(self, padding=1, **kwargs):
        super().__init__(**kwargs)
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)

    def call(self, inputs):
        return Tensor(torch.nn.functional.pad(C._raw(inputs), (0, 0, self.padding[0], self.padding[1])))


class MaxPool2D(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding="valid", data_format=None, **kwargs):
        super().__init__(**kwargs)
        self.pool_size = _tuple2(pool_size)
        self.strides = _tuple2(strides if strides is not None else pool_size)
        self.padding = padding.lower()
        assert self.pool_size[0] == self.pool_size[1] and self.strides[0] == self.strides[1]

    def call(self, inputs):
        return Tensor(_ops.max_pool2d(C._raw(inputs), self.pool_size[0], self.strides[0], self.padding))


MaxPooling2D = MaxPool2D


class AveragePooling2D(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding="valid", data_format=None, **kwargs):
        super().__init__(**kwargs)
        self.pool_size = _tuple2(pool_size)
        self.strides = _tuple2(strides if strides is not None else pool_size)
        self.padding = padding.lower()
        assert self.pool_size[0] == self.pool_size[1] and self.strides[0] == self.strides[1]

    def call(self, inputs):
        fn = _ops.avg_pool2d_same if self.padding == "same" else _ops.avg_pool2d_valid
        return Tensor(fn(C._raw(inputs), self.pool_size[0], self.strides[0]))


AvgPool2D = AveragePooling2D


class GlobalAveragePooling2D(Layer):
    def __init__(self, data_format=None, keepdims=False, **kwargs):
        super().__init__(**kwargs)
        self.keepdims = keepdims

    def call(self, inputs):
        return Tensor(C._raw(inputs).mean(dim=(1, 2), keepdim=self.keepdims))


class GlobalMaxPool2D(Layer):
    def __init__(self, data_format=None, keepdims=False, **kwargs):
        super().__init__(**kwargs)
        self.keepdims = keepdims

    def call(self, inputs):
        return Tensor(torch.amax(C._raw(inputs), dim=(1, 2), keepdim=self.keepdims))


GlobalMaxPooling2D = GlobalMaxPool2D


class GlobalAveragePooling1D(Layer):
    def __init__(self, data_format=None, keepdims=False, **kwargs):
        super().__init__(**kwargs)
        self.keepdims = keepdims

    def call(self, inputs, mask=None):
        return Tensor(C._raw(inputs).mean(dim=1, keepdim=self.keepdims))


class Flatten(Layer):
    def call(self, inputs):
        t = C._raw(inputs)
        return Tensor(t.reshape(t.shape[0], -1))


class LayerNormalization(Layer):
    def __init__(self, axis=-1, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros",
                 gamma_initializer="ones", **kwargs):
        super().__init__(**kwargs)
        assert axis == -1 and center and scale
        self.epsilon = epsilon
        self.beta_initializer, self.gamma_initializer = beta_initializer, gamma_initializer
        self.gamma = self.beta = None

    def build(self, input_shape):
        d = input_shape[-1]
        self.gamma = self.add_weight("gamma", shape=(d,), initializer=self.gamma_initializer)
        self.beta = self.add_weight("beta", shape=(d,), initializer=self.beta_initializer)

    def call(self, inputs):
        return Tensor(_ops.layer_norm(C._raw(inputs), self.gamma._t, self.beta._t, self.epsilon))


class BatchNormalization(Layer):
    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros",
                 gamma_initializer="ones", moving_mean_initializer="zeros", moving_variance_initializer="ones",
                 **kwargs):
        super().__init__(**kwargs)
        assert axis in (-1, 3) and center and scale
        self.momentum, self.epsilon = momentum, epsilon
        self.beta_initializer, self.gamma_initializer = beta_initializer, gamma_initializer
        self.moving_mean_initializer = moving_mean_initializer
        self.moving_variance_initializer = moving_variance_initializer
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None

    def build(self, input_shape):
        d = input_shape[-1]
        self.gamma = self.add_weight("gamma", shape=(d,), initializer=self.gamma_initializer)
        self.beta = self.add_weight("beta", shape=(d,), initializer=self.beta_initializer)
        self.moving_mean = self.add_weight("moving_mean", shape=(d,), initializer=self.moving_mean_initializer,
                                           trainable=False)
        self.moving_variance = self.add_weight("moving_variance", shape=(d,),
                                               initializer=self.moving_variance_initializer, trainable=False)

    def call(self, inputs, training=None):
        if training:
            raise NotImplementedError("the stand-in implements the inference path only")
        return Tensor(_ops.batch_norm(C._raw(inputs), self.gamma._t, self.beta._t, self.moving_mean._t,
                                      self.moving_variance._t, self.epsilon))


class InputLayer(Layer):
    pass


def __getattr__(name):
    """Layers the forward path never instantiates (bases of out-of-scope reference modules):
    importable, unusable."""
    if name.startswith("__"):
        raise AttributeError(name)

    def _init(self, *a, **k):
        raise NotImplementedError(f"tf.keras.layers.{name} is not provided by the stand-in")
    return type(name, (Layer,), {"__init__": _init})


__all__ = [n for n, v in list(globals().items()) if isinstance(v, type) and issubclass(v, Layer)] + ["OrderedDict"]
