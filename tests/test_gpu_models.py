"""pytest -m gpu: whole-model parity, HIP engine vs fp32 CPU oracle (helpers in model_checks.py)."""
import json
import os

import numpy as np
import pytest

import model_checks as mc
import test_architectures  # noqa: F401  (registers the miniature configs)

pytestmark = pytest.mark.gpu

MINIS = ["vit_test_model", "deit_test_model", "vit_hd64_test_model", "vit_hd80_test_model", "resnet_test_model_1", "resnet_test_model_2",
         "resnet50_mini_test_model", "seresnet_test_model", "swin_test_model", "swin_shift_test_model",
         "efficientnet_test_model", "efficientnet_same_test_model", "convnext_odd_test_model", "convnext_wide_test_model",
         "cait_hd48_test_model", "cait_hd32_test_model", "resnetd_test_model", "resnext_test_model", "resnext_wide_test_model",
         "ecaresnet_test_model", "resnetd_odd_test_model", "resnet_gn_test_model", "resnetblur_test_model",
         "resnetblur_basic_test_model"]
FULL = [("vit_tiny_patch16_224", 2), ("deit_tiny_distilled_patch16_224", 2), ("resnet18", 2), ("resnet50", 2),
        ("vit_base_patch16_224", 1), ("swin_tiny_patch4_window7_224", 2), ("efficientnet_b0", 2),
        ("swin_base_patch4_window7_224", 1), ("efficientnet_b4", 1), ("efficientnet_v2_b0", 2), ("mobilenet_v2_100", 2), ("convnext_tiny", 2),
        ("convnext_base_384_in22ft1k", 1), ("cait_xxs24_224", 2), ("cait_s24_224", 1), ("cait_m36_384", 1), ("resnet50d", 2),
        ("seresnet152d", 1), ("resnext50_32x4d", 1), ("ecaresnet50d", 1), ("resnet50_gn", 1), ("resnetblur50", 1)]


@pytest.mark.parametrize("name", MINIS)
def test_mini_model(name):
    r = mc.compare_model(name, batch=3)
    assert r["logits"] <= mc.TOL_LOGITS, r


@pytest.mark.parametrize("name,batch", FULL)
def test_full_model(name, batch):
    r = mc.compare_model(name, batch=batch)
    assert r["logits"] <= mc.TOL_LOGITS, r
    assert r["top1_agree"] == 1.0, r


# Configurations of the in-scope families that had no -m gpu test of their own (round-4 verdict): the window-12 / 384-pixel
# Swin (swin.py:551), the deepest EfficientNetV2 below XL (efficientnet.py:1420; 4.7e-2 in the all-configuration sweep: close
# to the bar, so the argmax statement is the one restricted to margins outside the row's own error) and a 32x8d ResNeXt-101
# (one tfimm_hip_gemm per group on the channel slice).
LARGE = [("swin_base_patch4_window12_384", 1), ("efficientnet_v2_l", 1), ("resnext101_32x8d", 1)]


@pytest.mark.parametrize("name,batch", LARGE)
def test_large_configurations(name, batch):
    r = mc.compare_model(name, batch=batch)
    assert r["logits"] <= mc.TOL_LOGITS, r
    assert r["top1_agree_outside_error_band"] == 1.0, r


# The two configurations of the 196 that the general bar does not fit (round-5 sweep: 0.25 / 0.32 rel-to-max): EfficientNetV2-XL
# (efficientnet.py:1512), 100 blocks deep, amplifies ANY bf16 rounding under random-init weights -- the fp32 oracle itself moves
# by 0.31 / 0.49 when only its kernels and input are rounded to bf16.  They are held to that stated, per-config bar
# (tests/golden/bf16_bars.json "deep_configs", tools/make_bf16_sensitivity.py: a property of model + weights, computed without
# the engine), and the float32 path of the same layer program to the reference's own 1e-3 -- which is the statement about
# the arithmetic.
_DEEP = {k: v for k, v in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_bars.json")))
         ["deep_configs"].items() if not k.startswith("_")}


@pytest.mark.parametrize("name", sorted(_DEEP))
def test_deep_configurations_against_their_stated_bars(name):
    from tfimm.engine import precision
    bar = _DEEP[name]
    r = mc.compare_model(name, batch=bar["batch"])
    assert r["logits"] <= bar["logits_bar"], (r, bar)
    assert r["top1_agree_outside_error_band"] == 1.0, r
    with precision.use("fp32"):
        r32 = mc.compare_model(name, batch=bar["batch"])
    assert r32["logits"] <= 1e-3 and r32["top1_agree"] == 1.0, r32


def test_plumbing_vit_tiny_b1():
    """BASELINE.json configs[0]: vit_tiny_patch16_224, batch 1."""
    r = mc.compare_model("vit_tiny_patch16_224", batch=1)
    assert r["shape"] == (1, 1000) and r["logits"] <= mc.TOL_LOGITS, r


def test_resnet_other_input_size():
    """convnets accept any spatial size at inference."""
    r = mc.compare_model("resnet18", batch=2, size=(160, 128))
    assert r["logits"] <= mc.TOL_LOGITS, r


def test_resnet_d_at_an_odd_input_size():
    """ResNet-D shortcuts at odd feature-map sizes (resnet.py:299-301): 250 -> 125 -> 63 -> 32 -> 16 -> 8."""
    r = mc.compare_model("resnet26d", batch=2, size=(250, 250))
    assert r["logits"] <= mc.TOL_LOGITS and r["top1_agree_outside_error_band"] == 1.0, r


@pytest.mark.parametrize("name", ["resnet50_mini_test_model", "seresnet_test_model", "swin_shift_test_model",
                                  "efficientnet_same_test_model", "resnet_gn_test_model", "resnetblur_test_model"])
def test_features_of_every_family(name):
    """return_features=True against the oracle for the families whose features had no GPU comparison
    (resnet.py:562-584, swin.py:467-517, efficientnet.py:270-345)."""
    r = mc.compare_model(name, batch=2, features=True)
    feats = [k for k in r if k.startswith("feat:")]
    assert len(feats) >= 4
    bad = {k: v for k, v in r.items() if (k.startswith("feat:") or k == "logits") and v > mc.TOL_LOGITS}
    assert not bad, bad


@pytest.mark.parametrize("name,batch", [("resnet50", 2), ("swin_tiny_patch4_window7_224", 1), ("efficientnet_b0", 2)])
def test_features_full_size(name, batch):
    r = mc.compare_model(name, batch=batch, features=True)
    bad = {k: v for k, v in r.items() if (k.startswith("feat:") or k == "logits") and v > mc.TOL_LOGITS}
    assert not bad, bad


def test_convnext_other_input_size_and_features():
    r = mc.compare_model("convnext_wide_test_model", batch=2, size=(40, 64), features=True)
    bad = {k: v for k, v in r.items() if (k.startswith("feat:") or k == "logits") and v > mc.TOL_LOGITS}
    assert not bad, bad


def test_cait_features_and_reference_mini():
    """every returned feature of the hd-48 mini (incl. the in-place class-token blocks' snapshots); the
    reference's own 4-channel mini runs through the catch-all kernels and gets the looser bar (see
    test_features_vit_mini)."""
    r = mc.compare_model("cait_hd48_test_model", batch=2, features=True)
    bad = {k: v for k, v in r.items() if (k.startswith("feat:") or k == "logits") and v > mc.TOL_LOGITS}
    assert not bad, bad
    r = mc.compare_model("cait_test_model", batch=3)
    assert r["logits"] <= 2 * mc.TOL_LOGITS, r


@pytest.mark.parametrize("name,size", [("vit_hd64_test_model", (80, 48)), ("cait_hd48_test_model", (48, 96)),
                                       ("vit_tiny_patch16_224", (160, 256))])
def test_interpolate_input(name, size):
    """ViT / CaiT at a non-native input size: position embeddings resized on the host (vit.py:433-442)."""
    r = mc.compare_model(name, batch=2, size=size, interpolate_input=True)
    assert r["logits"] <= mc.TOL_LOGITS, r


def test_graph_replay_matches_eager_launches():
    """model(x): first call launches eagerly, later calls replay a hipGraph of the same plan on a private input
    buffer -- same bits, also for new input values and after set_weights()."""
    import tfimm
    from tfimm.utils.init import synthetic_weights
    m = tfimm.create_model("resnet50_mini_test_model")
    m.set_weights(synthetic_weights(m))
    x1, x2 = mc.make_input(m.cfg, 3, seed=1), mc.make_input(m.cfg, 3, seed=2)
    eager1 = m(x1).numpy()                 # eager
    replay1 = m(x1).numpy()                # records, then replays
    replay2 = m(x2).numpy()
    replay1b = m(x1).numpy()
    assert m._captured, "second call should have recorded a graph"
    assert np.array_equal(eager1, replay1) and np.array_equal(eager1, replay1b)
    assert not np.array_equal(replay1, replay2)
    m.set_weights(synthetic_weights(m, 7))
    assert not m._captured
    again = m(x1).numpy()
    assert not np.array_equal(again, eager1)
    feats = m(x1, return_features=True)[1]      # another program (features): its own plan and recording
    assert "logits" in feats or len(feats) > 0


def test_micro_batch_equals_full_batch():
    import tfimm
    from tfimm.utils.init import synthetic_weights
    m = tfimm.create_model("resnet_test_model_2")
    m.set_weights(synthetic_weights(m))
    x = mc.make_input(m.cfg, 5)
    full = m(x).numpy()
    m.micro_batch = 2
    part = m(x).numpy()
    assert np.array_equal(full, part)


def test_features_vit_mini():
    """Intermediate features.  embed_dim = 4 (the reference's own mini) puts a LayerNorm over 4
    bf16-rounded values in front of every feature: one bf16 ulp of the residual stream moves a
    normalised value by several percent, so that model gets the looser bar; the 128-wide mini
    must meet the logits bar on every feature."""
    r = mc.compare_model("vit_hd64_test_model", batch=2, features=True)
    bad = {k: v for k, v in r.items() if k.startswith("feat:") and v > mc.TOL_LOGITS}
    assert not bad, bad
    r = mc.compare_model("vit_test_model", batch=2, features=True)
    bad = {k: v for k, v in r.items() if k.startswith("feat:") and v > 2 * mc.TOL_LOGITS}
    assert not bad, bad


@pytest.mark.parametrize("name", ["resnet50_mini_test_model", "vit_hd64_test_model", "convnext_odd_test_model"])
def test_deferred_uint8_preprocessing_matches_host_path(name):
    """create_preprocessing(defer=True): uint8 pixels go to the device and (v/255 - mean)/std runs inside the
    input-conversion kernel -- same logits, bit for bit, as preprocessing on the host in float32 (factory.py:165-167),
    eagerly launched and replayed from the recorded graph."""
    import tfimm
    from tfimm.utils.init import synthetic_weights
    m = tfimm.create_model(name)
    m.set_weights(synthetic_weights(m))
    H, W = m.cfg.input_size
    img = np.random.default_rng(5).integers(0, 256, (3, H, W, m.cfg.in_channels), dtype=np.uint8)
    host = m(tfimm.create_preprocessing(name)(img)).numpy()
    pre = tfimm.create_preprocessing(name, defer=True)
    eager = m(pre(img)).numpy()
    replay = m(pre(img)).numpy()
    img2 = 255 - img
    replay2 = m(pre(img2)).numpy()
    assert np.array_equal(host, eager) and np.array_equal(host, replay)
    assert np.array_equal(replay2, m(tfimm.create_preprocessing(name)(img2)).numpy())


def test_stem_reads_the_callers_image_like_the_padded_copy(monkeypatch):
    """ResNet's fused stem fed with the caller's float32 / bf16 RGB image (no separate conversion pass) gives the
    same logits, bit for bit, as the padded-copy path (TFIMM_NO_STEM_RAW=1) -- at the native and at another size."""
    import torch
    import tfimm
    from tfimm.utils.init import synthetic_weights
    m = tfimm.create_model("resnet18")
    m.set_weights(synthetic_weights(m))
    for size in (m.cfg.input_size, (160, 96)):
        prog = m.program(*size)
        plan = prog.make_plan(3)
        assert plan._stem_raw is not None
        x32 = torch.from_numpy(mc.make_input(m.cfg, 3, size=size)).cuda()
        for x in (x32, x32.to(torch.bfloat16)):
            monkeypatch.setenv("TFIMM_NO_STEM_RAW", "1")
            plan.run(x)
            a = plan.tensor_view(prog.outputs["logits"]).clone()
            monkeypatch.delenv("TFIMM_NO_STEM_RAW")
            plan.run(x)
            b = plan.tensor_view(prog.outputs["logits"]).clone()
            torch.cuda.synchronize()
            assert torch.equal(a, b)
