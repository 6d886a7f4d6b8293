"""create_preprocessing on the host (no GPU): the reference's formula and vector tiling (models/factory.py:128-171),
and the deferred uint8 wrapper that Model.__call__ hands to tfimm_hip_preprocess_input."""
import numpy as np
import pytest

import tfimm
import test_architectures  # noqa: F401  (registers the miniature configs)
from tfimm.models import DeferredInput


def _host(img, mean, std):
    x = img.astype(np.float32) / np.float32(255.0)
    return (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)


def test_formula_and_dtype():
    cfg = tfimm.models.registry.model_config("resnet50")
    img = np.random.default_rng(0).integers(0, 256, (2, 8, 8, 3), dtype=np.uint8)
    out = tfimm.create_preprocessing("resnet50")(img)
    assert out.dtype == np.float32 and np.array_equal(out, _host(img, cfg.mean, cfg.std))
    single = tfimm.create_preprocessing("resnet50")(img[0].astype(np.float32))      # single image, float input
    assert np.array_equal(single, out[0])
    assert tfimm.create_preprocessing("resnet50", dtype="float64")(img).dtype == np.float64


def test_mean_std_tiled_to_in_channels():
    cfg = tfimm.models.registry.model_config("resnet50")
    img = np.random.default_rng(1).integers(0, 256, (1, 4, 4, 5), dtype=np.uint8)
    out = tfimm.create_preprocessing("resnet50", in_channels=5)(img)
    mean = (list(cfg.mean) * 2)[:5]
    std = (list(cfg.std) * 2)[:5]
    assert np.array_equal(out, _host(img, mean, std))
    one = tfimm.create_preprocessing("resnet50", in_channels=1)(img[..., :1])
    assert np.array_equal(one, _host(img[..., :1], cfg.mean[:1], cfg.std[:1]))


def test_unknown_model():
    with pytest.raises(ValueError):
        tfimm.create_preprocessing("no_such_model")


def test_defer_wraps_uint8_only():
    import torch
    cfg = tfimm.models.registry.model_config("vit_tiny_patch16_224")
    pre = tfimm.create_preprocessing("vit_tiny_patch16_224", defer=True)
    img = np.random.default_rng(2).integers(0, 256, (2, 6, 6, 3), dtype=np.uint8)
    d = pre(img)
    assert isinstance(d, DeferredInput) and d.shape == (2, 6, 6, 3)
    assert d.mean == tuple(float(np.float32(v)) for v in cfg.mean)
    # what the kernel computes is what the host path computes
    assert np.array_equal(d.numpy(), tfimm.create_preprocessing("vit_tiny_patch16_224")(img))
    assert np.array_equal(np.asarray(d), d.numpy())
    dt = pre(torch.from_numpy(img))
    assert isinstance(dt, DeferredInput) and np.array_equal(dt.numpy(), d.numpy())
    # floats are preprocessed immediately, as without the flag
    f = pre(img.astype(np.float32))
    assert isinstance(f, np.ndarray) and np.array_equal(f, d.numpy())


def test_model_rejects_malformed_deferred_input():
    m = tfimm.create_model("vit_test_model")
    pre = tfimm.create_preprocessing("vit_test_model", defer=True)
    H, W = m.cfg.input_size
    with pytest.raises(ValueError):
        m(pre(np.zeros((H, W, m.cfg.in_channels), np.uint8)))                  # no batch dimension
    with pytest.raises(ValueError):
        m(pre(np.zeros((1, H, W, m.cfg.in_channels + 1), np.uint8)))           # channel mismatch
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="ROCm GPU"):                     # no CPU execution path
            m(pre(np.zeros((1, H, W, m.cfg.in_channels), np.uint8)))
