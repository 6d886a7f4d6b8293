"""The pin of the oracle to the reference.

* tests/golden/reference_weights.json -- variable names + shapes + feature names of reference
  models, dumped from ``model.weights`` of the reference's own classes (built over the stand-in
  TensorFlow, oracle/tools/make_reference_golden.py).  The engine's weight inventory
  (``weight_specs()``: what ``set_weights`` / timm ingestion key on) must equal it exactly; the
  build-time constants the reference never loads (swin ``attn_mask`` / ``relative_position_index``,
  swin.py:479-486) must be exactly the names the engine ignores on load.
* the stand-in itself (oracle/tf_shim): the Keras behaviours it restates -- lazy build, name scopes,
  Sequential's scope reset, ``training`` inheritance, variable tracking -- are unit-tested here.
* when /root/reference is present (build container only) the reference is re-run live and must
  reproduce the committed fixture bit for bit.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import test_architectures  # noqa: F401
import tfimm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "reference_weights.json")) as f:
    INVENTORY = json.load(f)


@pytest.mark.parametrize("name", sorted(INVENTORY))
def test_weight_inventory_equals_reference(name):
    ref = INVENTORY[name]
    model = tfimm.create_model(name)
    constants = set(ref["constants"])
    loadable = {k: tuple(v) for k, v in ref["variables"].items() if k not in constants}
    mine = {k: tuple(s.shape) for k, s in model._specs.items()}
    assert sorted(mine) == sorted(loadable)
    assert mine == loadable
    for k in constants:
        assert any(k.endswith(s) for s in model.keys_to_ignore_on_load), k
    assert list(model.feature_names) == ref["feature_names"]
    assert model.weight_names(with_prefix=True)[0] == f"{name}/{next(iter(model._specs))}:0"


# ---- every configuration: digests of the reference's loadable variables (names + shapes) and feature names ---------------
with open(os.path.join(ROOT, "tests", "golden", "reference_weight_digests.json")) as f:
    DIGESTS = json.load(f)


def _engine_inventories():
    """name -> (shapes, feature names) of every registered configuration, built without initialising any weight."""
    from tfimm.utils import init as winit
    fill, winit.initialize = winit.initialize, (lambda specs, mode="keras", seed=0: {})
    out = {}
    try:
        for name in DIGESTS:
            m = tfimm.models.model_class(name)(tfimm.models.model_config(name))
            out[name] = ({k: tuple(s.shape) for k, s in m._specs.items()}, list(m.feature_names))
    finally:
        winit.initialize = fill
    return out


def test_weight_inventory_digest_of_every_configuration_equals_reference():
    """oracle/tools/make_reference_golden.py --digests built every configuration the reference registers from the reference's
    own classes (over the stand-in TensorFlow), compared its variables name by name with the engine's and stored a sha256 of
    the loadable names + shapes + feature names.  The engine's inventories must still hash to the same values: what
    ``set_weights`` / timm ingestion key on is the reference's naming for all 196 configurations, not only the 24 whose full
    inventories are in reference_weights.json."""
    import hashlib
    registered = [n for n in tfimm.list_models() if not n.endswith("_test_model") and "_test_model_" not in n]
    assert sorted(DIGESTS) == sorted(registered), sorted(set(registered) ^ set(DIGESTS))
    assert len(DIGESTS) == 196
    bad = []
    for name, (shapes, feats) in _engine_inventories().items():
        text = "\n".join(f"{k} {tuple(int(d) for d in shapes[k])}" for k in sorted(shapes)) + "|" + ",".join(feats)
        ref = DIGESTS[name]
        if hashlib.sha256(text.encode()).hexdigest() != ref["digest"] or len(shapes) != ref["variables"]:
            bad.append(name)
        assert sum(int(np.prod(s)) for s in shapes.values()) == ref["parameters"], name
    assert not bad, bad


def test_inventory_covers_every_family_and_the_scored_models():
    for name in ("vit_base_patch16_224", "resnet50", "swin_base_patch4_window7_224", "efficientnet_b4", "cait_s24_224",
                 "convnext_tiny"):
        assert name in INVENTORY


# ---- the stand-in TensorFlow ----------------------------------------------------------------------
@pytest.fixture()
def tf():
    shim = os.path.join(ROOT, "oracle", "tf_shim")
    saved = {k: v for k, v in sys.modules.items() if k == "tensorflow" or k.startswith("tensorflow.")}
    sys.path.insert(0, shim)
    try:
        import tensorflow
        assert "standin" in tensorflow.__version__
        tensorflow.keras.backend.clear_session()
        yield tensorflow
    finally:
        sys.path.remove(shim)
        for k in list(sys.modules):
            if k == "tensorflow" or k.startswith("tensorflow."):
                del sys.modules[k]
        sys.modules.update(saved)


def test_shim_variable_names_follow_call_time_name_scopes(tf):
    L = tf.keras.layers

    class Inner(L.Layer):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.fc = L.Dense(3, name="fc")

        def build(self, input_shape):
            self.scale = self.add_weight("scale", shape=(input_shape[-1],), initializer="ones")

        def call(self, x):
            return self.fc(x * self.scale)

    class Outer(tf.keras.Model):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.blocks = [Inner(name=f"blocks/{j}") for j in range(2)]
            self.seq = tf.keras.Sequential([L.Dense(2, name="outer/seq/0"), L.BatchNormalization(name="outer/seq/1")])

        def call(self, x, training=False):
            for b in self.blocks:
                x = b(x)
            return self.seq(x, training=training)

    m = Outer(name="outer")
    y = m(np.ones((2, 5), np.float32))
    assert y.shape == (2, 2)
    names = [v.name for v in m.weights]
    # nested layers inherit the scopes of the calls they are built in ...
    assert "outer/blocks/0/scale:0" in names and "outer/blocks/1/fc/kernel:0" in names
    # ... layers of a Sequential do not (hence the reference's fully spelled names, resnet.py:309-330)
    assert "outer/seq/0/kernel:0" in names and "outer/seq/1/moving_variance:0" in names
    assert not any(n.startswith("outer/sequential") for n in names)
    assert len(names) == len(set(names)) == 2 * 3 + 2 + 4
    assert [v.name for v in m.non_trainable_weights] == ["outer/seq/1/moving_mean:0", "outer/seq/1/moving_variance:0"]


def test_shim_auto_names_and_training_inheritance(tf):
    L = tf.keras.layers
    seen = []

    class Probe(L.Layer):
        def call(self, x, training=None):
            seen.append(training)
            return x

    class Wrap(L.Layer):
        def __init__(self):
            super().__init__()
            self.p = Probe()

        def call(self, x, training=False):
            return self.p(x)            # not forwarded: Keras hands the enclosing call's value down

    a, b = L.Dense(1), L.Dense(1)
    assert (a.name, b.name) == ("dense", "dense_1")
    assert L.LayerNormalization().name == "layer_normalization" and L.Conv2D(1, 1).name == "conv2d"
    w = Wrap()
    w(np.zeros((1, 1), np.float32))
    w(np.zeros((1, 1), np.float32), training=True)
    assert seen == [False, True]
    assert L.Activation("linear")(np.zeros((1, 2), np.float32), training=False).shape == (1, 2)   # argument dropped


def test_shim_tensor_semantics(tf):
    x = tf.convert_to_tensor(np.arange(24, dtype=np.float64).reshape(2, 3, 4))
    assert x.dtype == tf.float64 and x.shape.as_list() == [2, 3, 4] and x.shape.ndims == 3
    b, n, c = tf.unstack(tf.shape(x))
    assert (b, n, c) == (2, 3, 4) and tf.reshape(x, (b, n * c)).shape == (2, 12)
    assert tf.where(x != 0, -100.0, x).dtype == tf.float64                      # swin.py:269-270
    r = tf.roll(tf.reshape(tf.range(5), (1, 5)), shift=(-2,), axis=[1]).numpy()   # out[i] = in[(i - shift) % n]
    assert r.tolist() == [[2, 3, 4, 0, 1]]
    q = tf.cast(x, tf.float32)
    assert np.allclose(tf.linalg.matmul(q, q, transpose_b=True).numpy(), np.einsum("bik,bjk->bij", x.numpy(), x.numpy()))
    assert tf.concat([tf.shape(x)[:-1], [2, 2]], axis=-1) == (2, 3, 2, 2)         # layers/norm.py:87
    p = tf.pad(tf.reshape(tf.range(4), (1, 4)), [[0, 0], [2, 1]], mode="REFLECT").numpy()
    assert p.tolist() == [[2, 1, 0, 1, 2, 3, 2]]
    t = tf.gather(tf.convert_to_tensor(np.arange(6.0).reshape(3, 2)), tf.convert_to_tensor(np.array([2, 0])))
    assert t.numpy().tolist() == [[4.0, 5.0], [0.0, 1.0]]


# ---- live re-run of the reference (build container only) ------------------------------------------
@pytest.mark.skipif(not os.path.isdir("/root/reference/tfimm"), reason="reference checkout not present")
def test_reference_rerun_reproduces_committed_fixture():
    names = ["vit_test_model", "resnet_test_model_2", "swin_shift_test_model", "efficientnet_same_test_model",
             "cait_hd48_test_model", "convnext_wide_test_model"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "tools", "make_reference_golden.py"), "--check"]
                       + names, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "identical to tests/golden/forward_golden.npz" in r.stdout


# ---- weight ingestion rules: name map + first-conv adaptation, pinned to the reference's own functions --
def test_name_map_of_every_variable_of_every_configuration_equals_the_reference_functions():
    """tests/golden/reference_name_map.json holds, per configuration, a digest over (variable, PyTorch key, transposition
    kind, layer name, weight name) computed by the REFERENCE's convert_tf_weight_name_to_pt_weight_name (utils/timm.py:39-106)
    and _get_layer_name / _get_weight_name (models/factory.py:253-280), imported unmodified by
    oracle/tools/make_name_map_golden.py.  The engine's functions must produce the same table for all 93 708 variables."""
    import hashlib
    from tfimm.models.factory import _layer_name
    from tfimm.utils.timm import convert_tf_weight_name_to_pt_weight_name
    with open(os.path.join(ROOT, "tests", "golden", "reference_name_map.json")) as f:
        gold = json.load(f)
    inv = _engine_inventories()
    assert sorted(gold["digests"]) == sorted(inv) and len(inv) >= 196
    total = 0
    for name, (shapes, _) in inv.items():
        lines = []
        for k in sorted(shapes):
            key, kind = convert_tf_weight_name_to_pt_weight_name(f"{name}/{k}:0", tuple(shapes[k]))
            lines.append((k, key, kind, _layer_name(k), k))
        total += len(lines)
        if name in gold["tables"]:
            assert [list(t) for t in lines] == gold["tables"][name], name      # readable diff for one model per family
        assert hashlib.sha256("\n".join("|".join(t) for t in lines).encode()).hexdigest() == gold["digests"][name], name
    assert total > 90000


def test_first_conv_adaptation_equals_the_reference_function():
    """_transform_first_conv (models/factory.py:282-305) run by oracle/tools/make_name_map_golden.py on a seeded kernel for
    1 .. 8 input channels: sum for one channel, tile + rescale otherwise, biases untouched."""
    from tfimm.models.factory import _transform_first_conv
    g = np.load(os.path.join(ROOT, "tests", "golden", "first_conv_golden.npz"))
    for c in range(1, 9):
        got = _transform_first_conv(g["kernel"], c)
        assert got.shape == g[f"kernel_in{c}"].shape
        np.testing.assert_allclose(got, g[f"kernel_in{c}"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(_transform_first_conv(g["bias"], 4), g["bias_in4"])
