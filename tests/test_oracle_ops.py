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


# ---- known answers for the ops round 2 left to "engine == oracle" (two implementations by one author) ---------------------
def _keys(t, A=-0.5):
    """Keys cubic convolution kernel (the published definition behind tf.image.resize(method="bicubic"))."""
    t = abs(t)
    if t < 1:
        return ((A + 2) * t - (A + 3)) * t * t + 1
    if t < 2:
        return ((A * t - 5 * A) * t + 8 * A) * t - 4 * A
    return 0.0


def _bicubic_1d_analytic(v, n_out):
    """half-pixel centres, taps outside the image weigh 0, the rest renormalised -- from the kernel's definition, exact offsets"""
    n_in = len(v)
    out = []
    for o in range(n_out):
        loc = (o + 0.5) * n_in / n_out - 0.5
        base = math.floor(loc)
        taps = [(i, _keys(loc - i)) for i in range(base - 1, base + 3) if 0 <= i < n_in]
        tot = sum(w for _, w in taps)
        out.append(sum(v[i] * w for i, w in taps) / tot)
    return np.array(out)


def test_resize_bicubic_hand_computed_4_to_8_and_5_to_3():
    # 4 -> 8: source offsets are exactly 0.25 / 0.75, Keys weights 0.8671875, 0.2265625, -0.0703125, -0.0234375 (sum 1)
    assert abs(_keys(0.25) - 0.8671875) < 1e-12 and abs(_keys(0.75) - 0.2265625) < 1e-12
    assert abs(_keys(1.25) + 0.0703125) < 1e-12 and abs(_keys(1.75) + 0.0234375) < 1e-12
    v = np.array([1.0, 2.0, 3.0, 4.0], np.float32)
    got = O.resize_bicubic_tf(torch.from_numpy(v.reshape(1, 1, 4, 1)), (1, 8)).numpy().reshape(-1)
    # first output: taps at -2, -1 fall outside; (0.8671875 * 1 - 0.0703125 * 2) / 0.796875
    assert abs(got[0] - (0.8671875 * 1 - 0.0703125 * 2) / 0.796875) < 1e-6
    # interior outputs of a linear ramp are the ramp itself at the half-pixel source location (cubic convolution is exact on lines)
    np.testing.assert_allclose(got[3:5], [2.25, 2.75], atol=1e-6)                  # the two outputs whose four taps are inside
    np.testing.assert_allclose(got, _bicubic_1d_analytic(v, 8), atol=1e-6)
    # 5 -> 3 (no antialiasing: four taps at the scaled location): the middle output lands exactly on source pixel 2
    v5 = np.array([0.0, 1.0, 4.0, 9.0, 16.0], np.float32)
    got = O.resize_bicubic_tf(torch.from_numpy(v5.reshape(1, 5, 1, 1)), (3, 1)).numpy().reshape(-1)
    assert abs(got[1] - 4.0) < 1e-6
    np.testing.assert_allclose(got, _bicubic_1d_analytic(v5, 3), rtol=2e-3, atol=2e-3)     # offset 1/3 -> 341/1024 in the table


def test_resize_bicubic_upscale_agrees_with_pillow():
    """Pillow's BICUBIC (a = -0.5, half-pixel centres, support clipped to the image and renormalised) is the same filter when
    UPSCALING (no antialias widening) -- an implementation that shares nothing with oracle/ops.py.  Difference: TensorFlow
    reads the kernel from a 1024-entry table at the rounded offset."""
    from PIL import Image
    rng = np.random.default_rng(3)
    for (h, w), (oh, ow) in (((5, 7), (12, 9)), ((14, 14), (24, 24)), ((3, 4), (4, 11))):
        img = rng.standard_normal((h, w)).astype(np.float32)
        ref = np.asarray(Image.fromarray(img, mode="F").resize((ow, oh), Image.BICUBIC))
        got = O.resize_bicubic_tf(torch.from_numpy(img.reshape(1, h, w, 1)), (oh, ow)).numpy().reshape(oh, ow)
        assert np.abs(got - ref).max() <= 4e-3 * np.abs(ref).max(), ((h, w), (oh, ow), float(np.abs(got - ref).max()))


def test_group_norm_hand_computed():
    # one image, two pixels, four channels in two groups: group 0 = {1, 2, 5, 6} (mean 3.5, var 4.25), group 1 = {3, 4, 7, 8}
    x = torch.tensor([[[[1., 2., 3., 4.], [5., 6., 7., 8.]]]])
    gamma, beta = torch.tensor([1., 2., 1., 1.]), torch.tensor([0., 1., 0., 0.])
    y = O.group_norm(x, gamma, beta, 2, 0.0).numpy().reshape(2, 4)
    s = math.sqrt(4.25)
    np.testing.assert_allclose(y[:, 0], [(1 - 3.5) / s, (5 - 3.5) / s], rtol=1e-6)
    np.testing.assert_allclose(y[:, 1], [2 * (2 - 3.5) / s + 1, 2 * (6 - 3.5) / s + 1], rtol=1e-6)
    np.testing.assert_allclose(y[:, 2], [(3 - 5.5) / s, (7 - 5.5) / s], rtol=1e-6)
    # epsilon inside the square root: a constant group maps to beta
    yc = O.group_norm(torch.ones(1, 1, 3, 4), gamma, beta, 2, 1e-5).numpy()
    np.testing.assert_allclose(yc.reshape(3, 4), np.tile([0., 1., 0., 0.], (3, 1)), atol=1e-6)


def test_blur_pool_hand_computed():
    x = torch.arange(1., 10.).reshape(1, 3, 3, 1)
    y = O.blur_pool2d(x, 1).numpy().reshape(3, 3)              # pad 1, REFLECT: row -1 is row 1, column -1 is column 1
    assert abs(y[1, 1] - 80 / 16) < 1e-6                       # (1 + 4 + 3 + 8 + 20 + 12 + 7 + 16 + 9) / 16
    assert abs(y[0, 0] - 48 / 16) < 1e-6                       # window [[5, 4, 5], [2, 1, 2], [5, 4, 5]]
    y2 = O.blur_pool2d(torch.ones(1, 6, 6, 2), 2).numpy()      # stride 2: pad (3 + 2) // 2 - 1 = 1, 6 -> 3
    assert y2.shape == (1, 3, 3, 2) and np.allclose(y2, 1.0)


def test_conv2d_dilation_hand_computed():
    x = torch.arange(7.).reshape(1, 1, 7, 1)
    k = torch.ones(1, 3, 1, 1)
    y = O.conv2d(x, k, dilation=2).numpy().reshape(-1)         # taps at x, x + 2, x + 4
    np.testing.assert_allclose(y, [0 + 2 + 4, 1 + 3 + 5, 2 + 4 + 6])
    ys = O.conv2d(x, k, dilation=2, padding="same").numpy().reshape(-1)    # effective kernel 5: pad (2, 2)
    np.testing.assert_allclose(ys, [0 + 2, 1 + 3, 0 + 2 + 4, 1 + 3 + 5, 2 + 4 + 6, 3 + 5, 4 + 6])
    d = O.depthwise_conv2d(torch.arange(14.).reshape(1, 1, 7, 2), torch.ones(1, 3, 2, 1), dilation=2).numpy().reshape(3, 2)
    np.testing.assert_allclose(d, [[0 + 4 + 8, 1 + 5 + 9], [2 + 6 + 10, 3 + 7 + 11], [4 + 8 + 12, 5 + 9 + 13]])
