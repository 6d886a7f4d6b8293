"""Pins the CPU oracle's op restatements (oracle/ops.py) with known answers and independent
numpy formulations (CPU).  These are the TF/Keras semantics of SURVEY.md Appendix A."""
import math

import numpy as np
import torch

from oracle import ops as O


def test_same_padding_amounts_match_tf_formula():
    # EfficientNet-B4 cases listed in SURVEY.md App. C
    assert O.same_pad_amounts(380, 3, 2) == (0, 1)
    assert O.same_pad_amounts(190, 3, 2) == (0, 1)
    assert O.same_pad_amounts(95, 5, 2) == (2, 2)
    assert O.same_pad_amounts(48, 3, 2) == (0, 1)
    assert O.same_pad_amounts(24, 5, 2) == (1, 2)
    assert O.same_pad_amounts(24, 3, 1) == (1, 1) and O.same_pad_amounts(24, 5, 1) == (2, 2)
    from tfimm.utils.etc import same_padding
    for n in (7, 12, 95, 380):
        for k in (1, 3, 5, 7):
            for s in (1, 2):
                out, b, a = same_padding(n, k, s)
                assert (b, a) == O.same_pad_amounts(n, k, s) and out == -(-n // s)


def test_conv2d_is_cross_correlation_nhwc_hwio():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 6, 7, 3)).astype(np.float32)
    k = rng.standard_normal((3, 2, 3, 4)).astype(np.float32)
    y = O.conv2d(torch.from_numpy(x), torch.from_numpy(k), stride=2).numpy()
    ref = np.zeros((2, 2, 3, 4))
    for oy in range(2):
        for ox in range(3):
            patch = x[:, 2 * oy:2 * oy + 3, 2 * ox:2 * ox + 2, :]
            ref[:, oy, ox, :] = np.einsum("bhwc,hwco->bo", patch, k)
    assert np.allclose(y, ref, atol=1e-5)


def test_conv2d_same_stride2_puts_extra_pad_at_bottom_right():
    x = torch.zeros(1, 4, 4, 1)
    x[0, 3, 3, 0] = 1.0
    k = torch.zeros(3, 3, 1, 1)
    k[0, 0, 0, 0] = 1.0        # picks the top-left tap
    y = O.conv2d(x, k, stride=2, padding="same")
    # pad (0,1): output (1,1) window starts at input (2,2); its top-left tap is x[2,2]=0; output (1,1) of a
    # symmetric (1,1) padding would start at (1,1) instead.
    k2 = torch.zeros(3, 3, 1, 1)
    k2[1, 1, 0, 0] = 1.0       # centre tap -> x[3,3] under TF-same
    assert O.conv2d(x, k2, stride=2, padding="same")[0, 1, 1, 0] == 1.0
    assert y.abs().sum() == 0


def test_depthwise_matches_per_channel_conv():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 5, 5, 3)).astype(np.float32)
    k = rng.standard_normal((3, 3, 3, 1)).astype(np.float32)
    y = O.depthwise_conv2d(torch.from_numpy(x), torch.from_numpy(k), padding="same").numpy()
    for c in range(3):
        yc = O.conv2d(torch.from_numpy(x[..., c:c + 1]), torch.from_numpy(k[:, :, c:c + 1, :]), padding="same").numpy()
        assert np.allclose(y[..., c], yc[..., 0], atol=1e-5)


def test_layer_norm_population_variance():
    x = torch.tensor([[1.0, 2.0, 3.0, 4.0]])
    y = O.layer_norm(x, torch.ones(4), torch.zeros(4), 0.0)
    assert np.allclose(y.numpy(), (np.arange(1, 5) - 2.5) / math.sqrt(1.25), atol=1e-6)
    g, b = torch.tensor([2.0, 1, 1, 1]), torch.tensor([0.5, 0, 0, 0])
    y2 = O.layer_norm(x, g, b, 1e-6)
    assert abs(float(y2[0, 0]) - (2 * (1 - 2.5) / math.sqrt(1.25 + 1e-6) + 0.5)) < 1e-5


def test_batch_norm_inference_and_fold_identity():
    from tfimm.engine import pack
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 4, 4, 3)).astype(np.float32)
    k = rng.standard_normal((1, 1, 3, 5)).astype(np.float32)
    g, b = rng.uniform(0.5, 1.5, 5).astype(np.float32), rng.standard_normal(5).astype(np.float32)
    m, v = rng.standard_normal(5).astype(np.float32), rng.uniform(0.5, 1.5, 5).astype(np.float32)
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    ref = O.batch_norm(O.conv2d(t(x), t(k)), t(g), t(b), t(m), t(v), 1e-3).numpy()
    s, sh = pack.bn_scale_shift(g, b, m, v, 1e-3)
    folded = O.conv2d(t(x), t(k * s.reshape(1, 1, 1, 5))).numpy() + sh
    assert np.allclose(ref, folded, atol=1e-5)


def test_activations_known_values():
    x = torch.tensor([1.0, -1.0, 0.0, 7.0])
    assert abs(float(O.activation(x, "gelu")[0]) - 0.8413447) < 1e-6          # exact erf GELU
    assert abs(float(O.activation(x, "gelu")[1]) + 0.1586553) < 1e-6
    assert abs(float(O.activation(x, "swish")[0]) - 1 / (1 + math.exp(-1))) < 1e-6
    assert float(O.activation(x, "relu6")[3]) == 6.0 and float(O.activation(x, "linear")[1]) == -1.0


def test_zero_padded_maxpool_differs_from_minus_inf_padding_on_negatives():
    x = -torch.ones(1, 4, 4, 1)
    y = O.max_pool2d(O.zero_pad2d(x, 1), 3, 2)
    assert float(y[0, 0, 0, 0]) == 0.0          # the zero border wins (resnet.py:538-540 semantics)
    assert float(y[0, 1, 1, 0]) == -1.0


def test_avg_pool_same_counts_only_valid_elements():
    x = torch.arange(9.0).reshape(1, 3, 3, 1)
    y = O.avg_pool2d_same(x, 2, 2)
    assert y.shape == (1, 2, 2, 1)
    assert float(y[0, 0, 0, 0]) == (0 + 1 + 3 + 4) / 4 and float(y[0, 1, 1, 0]) == 8.0 and float(y[0, 0, 1, 0]) == (2 + 5) / 2


def test_roll_direction():
    x = torch.arange(5.0)
    assert O.roll(x, (-2,), (0,)).tolist() == [2, 3, 4, 0, 1]      # out[i] = in[(i + 2) mod n]


def test_bf16_bits_round_to_nearest_even_like_torch():
    from tfimm.engine import pack
    rng = np.random.default_rng(3)
    a = np.concatenate([rng.standard_normal(10000).astype(np.float32) * 100,
                        np.array([1.00390625, 1.01171875, 0.0, -0.0, 3.0e38, 1e-40], np.float32)])
    ours = pack.bf16_bits_to_f32(pack.to_bf16_bits(a))
    ref = torch.from_numpy(a).to(torch.bfloat16).float().numpy()
    assert np.array_equal(ours, ref)
