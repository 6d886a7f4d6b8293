"""Weight interchange: PyTorch/timm state_dict -> engine weights, position-embedding
resize, transfer_weights across input sizes.  CPU only."""
import logging

import numpy as np
import pytest
import torch

import test_architectures  # noqa: F401
import tfimm
from oracle import ops as oops
from tfimm.layers import interpolate_pos_embeddings, resize_bicubic
from tfimm.models.factory import transfer_weights
from tfimm.utils.init import synthetic_weights
from tfimm.utils.timm import convert_tf_weight_name_to_pt_weight_name, load_pytorch_weights_in_model


@pytest.mark.parametrize("tf_name,shape,pt_name,kind", [
    ("resnet50/layer1/0/conv1/kernel:0", (1, 1, 64, 64), "layer1.0.conv1.weight", "conv2d"),
    ("resnet50/layer1/0/bn1/moving_variance:0", (64,), "layer1.0.bn1.running_var", "no"),
    ("resnet50/layer1/0/bn1/moving_mean:0", (64,), "layer1.0.bn1.running_mean", "no"),
    ("resnet50/layer1/0/bn1/gamma:0", (64,), "layer1.0.bn1.weight", "no"),
    ("resnet50/layer1/0/bn1/beta:0", (64,), "layer1.0.bn1.bias", "no"),
    ("vit_base/blocks/3/attn/qkv/kernel:0", (768, 2304), "blocks.3.attn.qkv.weight", "simple"),
    ("vit_base/pos_embed:0", (1, 197, 768), "pos_embed", "no"),
    ("efficientnet_b0/blocks/1/0/conv_dw/depthwise_kernel:0", (3, 3, 96, 1), "blocks.1.0.conv_dw.weight", "conv2d"),
    ("convnext_tiny/stages/0/blocks/1/gamma:0", (96,), "stages.0.blocks.1.weight", "no"),
    ("m/head/remove/fc/kernel:0", (8, 4), "head.fc.weight", "simple"),         # auxiliary level dropped
    ("m/layers/tf_only___pt_name/bias:0", (4,), "layers.pt_name.bias", "no"),  # $1___$2 -> $2
    ("m/blocks_._0/bias:0", (4,), "blocks.0.bias", "no"),                       # _._ -> level separator
])
def test_name_and_layout_rules(tf_name, shape, pt_name, kind):
    """reference utils/timm.py:39-106, one case per rule."""
    assert convert_tf_weight_name_to_pt_weight_name(tf_name, shape) == (pt_name, kind)


def _to_pt_state_dict(model, weights):
    """The inverse map, written independently: what a timm checkpoint of this model would hold."""
    sd = {}
    for name, w in weights.items():
        parts = ("/" + name).replace("/remove/", "/")[1:].split("/")
        leaf = parts[-1]
        w = np.asarray(w, np.float32)
        if leaf == "depthwise_kernel":
            kh, kw, c, _ = w.shape
            t = np.transpose(w.reshape(kh, kw, 1, c), (3, 2, 0, 1))        # (C, 1, kh, kw)
        elif leaf == "kernel" and w.ndim == 4:
            t = np.transpose(w, (3, 2, 0, 1))                               # HWIO -> OIHW
        elif leaf == "kernel":
            t = w.T                                                         # (in, out) -> (out, in)
        else:
            t = w
        leaf_pt = {"kernel": "weight", "depthwise_kernel": "weight", "gamma": "weight", "beta": "bias",
                   "moving_mean": "running_mean", "moving_variance": "running_var"}.get(leaf, leaf)
        key = ".".join(parts[:-1] + [leaf_pt])
        if leaf == "gamma" and len(parts) >= 2 and parts[-2].isdigit():
            key = ".".join(parts[:-1] + ["gamma"])                          # ConvNeXt LayerScale is "....gamma" in timm
        sd[key] = torch.from_numpy(np.ascontiguousarray(t))
        if leaf == "moving_mean":
            sd[".".join(parts[:-1] + ["num_batches_tracked"])] = torch.tensor(7)
    return sd


@pytest.mark.parametrize("name", ["vit_test_model", "deit_test_model", "resnet_test_model_2", "seresnet_test_model",
                                  "swin_test_model", "efficientnet_test_model", "convnext_test_model", "cait_test_model"])
def test_state_dict_round_trip(name, caplog):
    src = tfimm.create_model(name)
    w = synthetic_weights(src, 7)
    sd = _to_pt_state_dict(src, w)
    dst = tfimm.create_model(name)
    with caplog.at_level(logging.WARNING):
        load_pytorch_weights_in_model(dst, sd)
    assert not [r for r in caplog.records if "were not used" in r.getMessage()], "num_batches_tracked must be ignored silently"
    for k, v in w.items():
        assert dst.weights[k].shape == v.shape and np.array_equal(dst.weights[k], v), k


def test_state_dict_missing_and_unexpected_keys(caplog):
    m = tfimm.create_model("vit_test_model")
    sd = _to_pt_state_dict(m, synthetic_weights(m, 3))
    del sd["blocks.0.attn.proj.bias"]
    with pytest.raises(AttributeError, match="blocks.0.attn.proj.bias"):
        load_pytorch_weights_in_model(tfimm.create_model("vit_test_model"), sd)
    sd["extra.weight"] = torch.zeros(2)
    with caplog.at_level(logging.WARNING):
        load_pytorch_weights_in_model(tfimm.create_model("vit_test_model"), sd, allow_missing_keys=True)
    text = " ".join(r.getMessage() for r in caplog.records)
    assert "extra.weight" in text and "blocks.0.attn.proj.bias" in text


def test_bicubic_resize_known_answers():
    """tf.image.resize(bicubic) semantics: half-pixel centres, Keys A=-0.5, taps outside
    the image dropped and the rest renormalised."""
    const = np.full((1, 5, 4, 2), 3.0, np.float32)
    assert np.abs(resize_bicubic(const, (11, 9)) - 3.0).max() < 1e-6
    x = np.random.default_rng(0).standard_normal((1, 6, 7, 3)).astype(np.float32)
    assert np.array_equal(resize_bicubic(x, (6, 7)), x)                      # same size: exact copy
    ramp = np.arange(10, dtype=np.float32).reshape(1, 10, 1, 1)
    up = resize_bicubic(ramp, (20, 1))[0, :, 0, 0]
    assert np.allclose(up[4:16], np.arange(4, 16) * 0.5 - 0.25, atol=1e-5)  # cubic convolution is exact on lines
    # hand-computed: [0, 1] -> 4 samples; first output sits at -0.25: taps 0 and 1 with Keys weights
    # 0.8671875 and -0.0703125, renormalised
    two = np.array([0.0, 1.0], np.float32).reshape(1, 2, 1, 1)
    got = resize_bicubic(two, (4, 1))[0, :, 0, 0]
    w1 = -0.0703125 / (0.8671875 - 0.0703125)
    assert abs(got[0] - w1) < 1e-6 and abs(got[3] - (1 - w1)) < 1e-6 and abs(got[1] + got[2] - 1.0) < 1e-6


@pytest.mark.parametrize("size", [(24, 24), (7, 9), (16, 12), (3, 5)])
def test_bicubic_two_implementations_agree(size):
    x = np.random.default_rng(1).standard_normal((1, 14, 14, 8)).astype(np.float32)
    assert np.abs(resize_bicubic(x, size) - oops.resize_bicubic_tf(x, size).numpy()).max() < 1e-6


def test_transfer_weights_resizes_pos_embed():
    """factory.py:174-250 + vit.py:414-420: a model created at another input size gets interpolated embeddings;
    class-token rows are carried over."""
    src = tfimm.create_model("vit_test_model")
    src.set_weights(synthetic_weights(src, 5))
    dst = tfimm.create_model("vit_test_model", input_size=(48, 64))
    transfer_weights(src, dst)
    want = interpolate_pos_embeddings(src.weights["pos_embed"], (4, 4), (6, 8), 1)
    assert dst.weights["pos_embed"].shape == (1, 49, 4)
    assert np.array_equal(dst.weights["pos_embed"], want)
    assert np.array_equal(dst.weights["pos_embed"][:, :1], src.weights["pos_embed"][:, :1])
    assert np.array_equal(dst.weights["blocks/0/attn/qkv/kernel"], src.weights["blocks/0/attn/qkv/kernel"])
    c_src = tfimm.create_model("cait_test_model")
    c_src.set_weights(synthetic_weights(c_src, 6))
    c_dst = tfimm.create_model("cait_test_model", input_size=(64, 32))
    transfer_weights(c_src, c_dst)
    assert c_dst.weights["pos_embed"].shape == (1, 32, 4)


def test_interpolate_input_oracle_and_lowering():
    """interpolate_input=True: the oracle follows vit.py:433-442 / cait.py:407-415 and the engine lowers a program
    for the other size (no GPU needed to build it); without the flag the size mismatch is an error."""
    import oracle
    for name, sz in (("vit_hd64_test_model", (80, 48)), ("cait_hd48_test_model", (48, 96))):
        m = tfimm.create_model(name, interpolate_input=True)
        w = synthetic_weights(m, 11)
        m.set_weights(w)
        x = np.random.default_rng(2).standard_normal((2, *sz, 3)).astype(np.float32)
        y = oracle.forward(m.cfg, w, x)
        assert y.shape == (2, 10) and np.isfinite(y).all()
        prog = m.program(*sz)
        assert prog.outputs["logits"].C == 10
        strict = tfimm.create_model(name)
        strict.set_weights(w)
        with pytest.raises(ValueError, match="interpolate_input"):
            strict.program(*sz)


def test_strict_loading_reports_missing_and_leaves_the_model_untouched(tmp_path):
    """a truncated checkpoint must fail loudly (the reference's Keras loader does); a mis-shaped one must not leave
    the model half-updated"""
    import pytest
    import test_architectures  # noqa: F401
    import tfimm
    from tfimm.utils.init import synthetic_weights
    m = tfimm.create_model("resnet_test_model_1")
    w = synthetic_weights(m, 3)
    m.set_weights(w)
    before = {k: v.copy() for k, v in m.weights.items()}
    dropped = next(iter(w))
    partial = {k: v for k, v in w.items() if k != dropped}
    with pytest.raises(KeyError, match="missing"):
        m.set_weights(partial)
    np.savez(tmp_path / "partial.npz", **partial)
    with pytest.raises(KeyError, match="missing"):
        m.load_weights(str(tmp_path / "partial.npz"))
    m.set_weights(partial, strict=False)                 # explicit opt-out still works
    bad = dict(synthetic_weights(m, 4))
    last = list(bad)[-1]
    bad[last] = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError, match="shape"):
        m.set_weights(bad)
    assert all(np.array_equal(m.weights[k], before[k]) for k in before)      # nothing of `bad` was committed
