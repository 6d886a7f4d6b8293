"""Layer-program lowering decisions that can be checked without a GPU: which models get the fused ResNet stem
(tfimm_hip_stem_conv_pool) and which keep convolution + max pooling as two ops (engine/graph.py, Builder.conv)."""
import pytest

import test_architectures  # noqa: F401  (registers the miniature configs)
import tfimm
from tfimm.utils.init import synthetic_weights


def _kinds(name, size=None, want_features=False, **kw):
    m = tfimm.create_model(name, **kw)
    m.set_weights(synthetic_weights(m))
    prog = m.program(*(size or m.cfg.input_size), **({"want_features": True} if want_features else {}))
    return [op.kind for op in prog.ops], prog


def test_plain_resnet_stem_is_one_op():
    kinds, prog = _kinds("resnet50")
    assert kinds[:3] == ["cast_input", "stem_pool", "gemm"] and "maxpool" not in kinds
    stem = prog.ops[1]
    # resnet.py:505-512, 538-540: 224 -> conv 112 -> pool 56; the padded pair view the kernel expects
    assert (stem.attrs["OH"], stem.attrs["OW"], stem.attrs["Hp"], stem.attrs["Wp2"]) == (112, 112, 229, 115)
    assert prog.tensors[stem.output].H == 56 and prog.tensors[stem.output].C == 64
    # the convolution's flops still count (bench.py's model_tflops / gflops_per_image)
    assert abs(prog.flops_per_image() / 1e9 - 8.178) < 0.01


def test_other_sizes_and_the_112_column_limit():
    kinds, prog = _kinds("resnet18", size=(160, 128))
    assert kinds[1] == "stem_pool" and (prog.ops[1].attrs["OH"], prog.ops[1].attrs["OW"]) == (80, 64)
    kinds, _ = _kinds("resnet18", size=(224, 448))         # 224 output columns: two ops
    assert kinds[1:3] == ["gemm", "maxpool"]


@pytest.mark.parametrize("name", ["resnet26d", "resnet26t", "resnetrs50"])
def test_deep_stems_stay_unfused_and_avg_down_is_one_folded_conv(name):
    """ResNet-D shortcuts (AveragePooling2D(2, 2, 'same') + 1x1 conv, resnet.py:295-312) lower to ONE 2x2 stride-2
    convolution whose taps are the 1x1 kernel / 4; deep stems never take the fused 7x7 stem kernel."""
    kinds, prog = _kinds(name)
    assert "stem_pool" not in kinds and kinds.count("maxpool") <= 1
    # round 6: the four taps are a 2 x 2 window of the SECOND operand of the block's last convolution (tfimm_gemm_desc::a2_window)
    # -- except behind an SE gate (resnetrs50), where the shortcut stays the folded 2 x 2 convolution of its own
    own = [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("KH") == 2 and op.attrs.get("stride") == 2]
    duals = [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("dual")]
    assert (len(own), len(duals)) == ((3, 0) if name == "resnetrs50" else (0, 3))      # the three strided stage transitions
    for op in own:
        assert op.attrs["K"] == 4 * op.attrs["Cin"] and op.attrs["OH"] * 2 == op.attrs["H"]
    for op in duals:
        du = op.attrs["dual"]
        assert du["window"] == 2 and du["stride"] == 2 and (du["H"], du["W"]) == (2 * du["OH"], 2 * du["OW"])
        assert op.attrs["ldw"] == -(-op.attrs["K"] // 64) * 64 + 4 * (-(-du["K2"] // 64) * 64)


@pytest.mark.parametrize("name", ["resnet26d"])
def test_avg_down_as_its_own_folded_conv_when_the_fold_is_off(name, monkeypatch):
    monkeypatch.setenv("TFIMM_NO_FOLD_SHORTCUT", "1")
    kinds, prog = _kinds(name)
    folded = [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("KH") == 2 and op.attrs.get("stride") == 2]
    assert len(folded) == 3
    for op in folded:
        assert op.attrs["K"] == 4 * op.attrs["Cin"] and op.attrs["OH"] * 2 == op.attrs["H"]


def test_avg_down_at_odd_sizes_pools_explicitly():
    """odd feature maps: AveragePooling2D(2, 2, 'same') clips its last window (resnet.py:299-301), so the shortcut is
    tfimm_hip_avg_pool + the 1x1 convolution instead of the folded 2x2 one; convnets accept any input size"""
    m = tfimm.create_model("resnet26d")
    m.set_weights(synthetic_weights(m))
    prog = m.program(200, 200)           # 200 -> stem 100 -> pool 50 -> 25 (odd) -> 13 (odd) -> 7
    pools = [op for op in prog.ops if op.kind == "avg_pool"]
    assert [(op.attrs["H"], prog.tensors[op.output].H) for op in pools] == [(25, 13), (13, 7)]
    folded = [op for op in prog.ops if op.kind == "gemm" and (op.attrs.get("dual") or {}).get("window") == 2]
    assert len(folded) == 1 and folded[0].attrs["dual"]["H"] == 50


def test_environment_switch_keeps_the_two_op_path(monkeypatch):
    monkeypatch.setenv("TFIMM_NO_STEM_FUSION", "1")
    kinds, _ = _kinds("resnet18")
    assert kinds[1:3] == ["gemm", "maxpool"]


def test_non_rgb_input_keeps_the_conversion_pass():
    """in_channels = 1: the stem is still fused (4 stored channels), but only RGB images are read raw by it."""
    kinds, prog = _kinds("resnet18", in_channels=1)
    assert kinds[1] == "stem_pool" and prog.input_shape[2] == 1


def test_avg_pool_then_1x1_conv_equals_the_folded_2x2_conv():
    """the identity the ResNet-D lowering relies on, checked with the oracle's TF-semantics ops"""
    import numpy as np
    import torch
    from oracle import ops
    r = np.random.default_rng(3)
    x = torch.from_numpy(r.standard_normal((2, 8, 6, 5)).astype(np.float32))
    w = r.standard_normal((1, 1, 5, 7)).astype(np.float32)
    ref = ops.conv2d(ops.avg_pool2d_same(x, 2, 2), torch.from_numpy(w))
    folded = ops.conv2d(x, torch.from_numpy(np.tile(w * 0.25, (2, 2, 1, 1))), stride=2)
    assert ref.shape == folded.shape == (2, 4, 3, 7)
    assert float((ref - folded).abs().max()) < 1e-5


def test_oracle_runs_the_resnet_d_mini():
    import numpy as np
    import model_checks as mc
    import oracle
    m = tfimm.create_model("resnetd_test_model")
    w = synthetic_weights(m)
    y = oracle.forward(m.cfg, w, mc.make_input(m.cfg, 2))
    assert y.shape == (2, 10) and np.isfinite(y).all()


def test_resnext_and_eca_lower_to_existing_ops_and_the_oracle_runs_them():
    import numpy as np
    import model_checks as mc
    import oracle
    kinds, prog = _kinds("resnext_test_model")
    assert kinds[1] == "stem_pool"
    # widths 16 / 32 / 48 / 64 in 4 groups: 32 and 64 run on 32-channel super-groups (tfimm_hip_grouped_conv3x3); 16 is
    # not a whole super-group and 48 has 12-channel groups (32 % 12 != 0): those take the dense block-diagonal expansion,
    # whose algorithmic K stays 9 * Cin / groups
    grouped = [op for op in prog.ops if op.kind == "grouped_conv"]
    dense = [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("KH") == 3]
    assert [op.attrs["C"] for op in grouped] == [32, 64] and [op.attrs["Cin"] for op in dense] == [16, 48]
    assert all(op.attrs["K_true"] * 4 == 9 * op.attrs["Cin"] for op in dense)
    assert all(op.attrs["flops"] == 2 * prog.tensors[op.output].rows * op.attrs["C"] * 9 * (op.attrs["C"] // 4) for op in grouped)
    kinds, _ = _kinds("ecaresnet_test_model")
    assert kinds.count("scale_channels") == 4 and kinds.count("mean_rows") == 5     # 4 gates + the head pooling
    for name in ("resnext_test_model", "ecaresnet_test_model"):
        m = tfimm.create_model(name)
        y = oracle.forward(m.cfg, synthetic_weights(m), mc.make_input(m.cfg, 2))
        assert y.shape == (2, 10) and np.isfinite(y).all()


def test_resnet50_stage1_tails_are_one_launch_each():
    """conv2 + bn2 + relu + conv3 + bn3 + shortcut + relu of the three stage-1 blocks lower to tfimm_hip_conv_chain (the
    64-channel intermediate never reaches HBM); the algorithmic FLOP count is unchanged; TFIMM_NO_CHAIN=1 keeps two GEMMs"""
    kinds, prog = _kinds("resnet50")
    chains = [op for op in prog.ops if op.kind == "conv_chain" and op.attrs["C1"] == 64]
    assert len(chains) == 3 and all((op.attrs["H"], op.attrs["C1"], op.attrs["N2"]) == (56, 64, 256) for op in chains)
    # ... and so do the three stride-1 blocks of stage 2 (128 -> 128 -> 512 at 28 x 28: csrc/conv_strip.hip, round 4)
    chains2 = [op for op in prog.ops if op.kind == "conv_chain" and op.attrs["C1"] == 128]
    assert [(op.attrs["H"], op.attrs["N2"], op.attrs["has_residual"], bool(op.attrs.get("has_ds"))) for op in chains2] == [(28, 512, True, False)] * 3
    # the first block's shortcut convolution (64 -> 256 on the stem output) is multiplied inside its tail: no launch, no tensor
    assert [(op.attrs["has_residual"], bool(op.attrs.get("has_ds"))) for op in chains] == [(False, True), (True, False), (True, False)]
    assert not [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("K") == 64 and op.attrs["N"] == 256]
    assert abs(prog.flops_per_image() / 1e9 - 8.178) < 0.01
    kinds, _ = _kinds("resnet50", size=(256, 256))          # rows of 64 pixels: wider than the kernel's input strip
    assert "conv_chain" not in kinds


def test_chain_switch(monkeypatch):
    monkeypatch.setenv("TFIMM_NO_CHAIN", "1")
    kinds, prog = _kinds("resnet50")
    assert "conv_chain" not in kinds and abs(prog.flops_per_image() / 1e9 - 8.178) < 0.01


def test_wide_groups_run_as_one_gemm_per_group_on_channel_slices():
    """groups of 64+ channels (ResNeXt-101 32x16d / 32d / 48d): neither the super-group kernel nor a block-diagonal dense
    expansion -- one implicit-GEMM launch per group that gathers its channel slice in place (pix_pitch) and writes its
    slice of the output (ldc, out_col)"""
    kinds, prog = _kinds("resnext_wide_test_model")
    grouped = [op for op in prog.ops if op.kind == "grouped_conv"]
    split = [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("pix_pitch")]
    assert [op.attrs["C"] for op in grouped] == [64]                       # 2 groups of 32: one super-group each
    assert [(op.attrs["Cin"], op.attrs["pix_pitch"], op.attrs["a_byte_offset"], op.attrs["out_col"], op.attrs["ldc"])
            for op in split] == [(64, 128, 0, 0, 128), (64, 128, 128, 64, 128), (96, 192, 0, 0, 192), (96, 192, 192, 96, 192),
                                 (128, 256, 0, 0, 256), (128, 256, 256, 128, 256)]
    assert all(op.attrs["K"] == 9 * op.attrs["Cin"] and op.attrs["N"] == op.attrs["Cin"] for op in split)
    assert len({op.output for op in split}) == 3                           # the two groups of a layer share one output tensor


def test_every_registered_resnet_is_supported():
    """all configurations the reference registers in its ResNet module lower (checked on the configurations:
    building all ~70 programs would take minutes)"""
    for name in tfimm.list_models(module="resnet"):
        tfimm.create_model(name).check_supported()


def test_group_norm_and_blur_pool_lowering():
    kinds, prog = _kinds("resnet_gn_test_model")
    # every convolution is followed by its own GroupNorm kernel; nothing is folded, the stem is not fused
    assert "stem_pool" not in kinds and kinds.count("group_norm") == 1 + 4 * 3 + 4
    last = [op for op in prog.ops if op.kind == "group_norm" and op.attrs["has_residual"]]
    assert len(last) == 4 and all(op.attrs["act_after"] == "relu" and op.attrs["act"] == "" for op in last)
    kinds, prog = _kinds("resnetblur_test_model")
    blurs = [op for op in prog.ops if op.kind == "blur_pool"]
    assert len(blurs) == 1 + 3 and "stem_pool" not in kinds
    assert [op.attrs["stride"] for op in prog.ops if op.kind == "maxpool"] == [1]         # resnet.py:534
    strided = [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("KH") == 3 and op.attrs.get("stride") == 2]
    assert not strided                     # the blur layer takes care of the stride (resnet.py:132, 233)


def test_programs_of_one_model_share_their_packed_constants():
    """a second program of the same model (other input size, feature variant) re-uses the uploaded constants: the cache is
    keyed by (name, shape, dtype, checksum), so only size-dependent constants are new (device = "cpu": no GPU needed)"""
    m = tfimm.create_model("resnet50_mini_test_model")
    m.set_weights(synthetic_weights(m))
    p1, p2, p3 = m.program(64, 64), m.program(96, 64), m.program(64, 64, want_features=True)
    p1.upload("cpu")
    n1 = len(m._const_cache)
    p2.upload("cpu")
    p3.upload("cpu")
    assert n1 > 10 and len(m._const_cache) == n1          # nothing size- or feature-dependent in a ResNet's constants
    assert all(a is b for a, b in zip(p1._dev_consts, p3._dev_consts))
    m.set_weights(synthetic_weights(m, 5))
    assert not m._const_cache


def test_narrow_mbconv_fronts_are_one_launch(monkeypatch):
    """expansion 1x1 + depthwise of the inverted-residual blocks whose input has at most 32 channels lower to
    tfimm_hip_expand_dwconv (the expanded tensor never reaches HBM); FLOP count and the following squeeze-excite are unchanged;
    TFIMM_NO_MBCONV_FUSION=1 keeps the GEMM + depthwise pair"""
    kinds, prog = _kinds("efficientnet_b0")
    stem = [op for op in prog.ops if op.kind == "expand_dwconv" and op.attrs.get("stem")]
    fused = [op for op in prog.ops if op.kind == "expand_dwconv" and not op.attrs.get("stem")]
    # conv_stem + bn1 + act + the first block's depthwise layer: the 3x3 / stride 2 convolution reads the zero-bordered image
    # (224 -> 225 rows: TF "same" pads one row / column at the bottom / right), its 112 x 112 x 32 output stays in LDS
    assert len(stem) == 1 and kinds[:2] == ["cast_input", "expand_dwconv"]
    a = stem[0].attrs
    assert (a["Cin"], a["C"], a["H"], a["W"], a["img_h"], a["img_w"], a["k"], a["stride"]) == (4, 32, 112, 112, 225, 225, 3, 1)
    assert [(op.attrs["Cin"], op.attrs["C"], op.attrs["k"], op.attrs["stride"], op.attrs["H"]) for op in fused] == \
        [(16, 96, 3, 2, 112), (24, 144, 3, 1, 56), (24, 144, 5, 2, 56)]
    assert all(op.attrs["sums"] is not None and op.attrs["Cpad"] % 32 == 0 for op in fused)
    for op in fused:       # TF "same" padding of the EXPANDED tensor: nothing on top / left at stride 2 on even sizes
        assert (op.attrs["pad_t"], op.attrs["pad_l"]) == ((1, 1) if op.attrs["stride"] == 1 else
                                                          ((op.attrs["k"] - 2) // 2,) * 2)
    flops = prog.flops_per_image()
    monkeypatch.setenv("TFIMM_NO_MBCONV_FUSION", "1")
    kinds2, prog2 = _kinds("efficientnet_b0")
    assert "expand_dwconv" not in kinds2 and len(kinds2) == len(kinds) + 4 and prog2.flops_per_image() == flops
    monkeypatch.delenv("TFIMM_NO_MBCONV_FUSION")
    kinds3, _ = _kinds("efficientnet_b0", want_features=True)          # the "stem" feature needs the stem output in HBM
    assert kinds3[:3] == ["cast_input", "gemm", "dwconv"]


def test_layernorms_with_one_dense_reader_are_folded(monkeypatch):
    """norm1 -> qkv and norm2 -> fc1 lower to a statistics pass + ONE GEMM over the raw rows (gamma in the weights, beta . W in
    the bias, column sums of the rounded weights as correction fragments); only the final norm stays a LayerNorm launch;
    TFIMM_NO_LN_FOLD=1 keeps the LayerNorm launches"""
    import numpy as np
    from tfimm.engine import pack
    kinds, prog = _kinds("vit_tiny_patch16_224")
    assert kinds.count("row_stats") == 24 and kinds.count("layernorm") == 1
    folded = [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("ln")]
    assert len(folded) == 24 and all(len(op.inputs) == 2 and "ln_c1" in op.consts for op in folded)
    op = folded[0]
    m = tfimm.create_model("vit_tiny_patch16_224")
    w = synthetic_weights(m)
    m.set_weights(w)
    g, bt = w["blocks/0/norm1/gamma"], w["blocks/0/norm1/beta"]
    k, b = w["blocks/0/attn/qkv/kernel"], w["blocks/0/attn/qkv/bias"]
    wt = prog.consts[op.consts["wt"]].host
    np.testing.assert_array_equal(wt[:, :192], pack.to_bf16_bits((k * g[:, None]).T))
    np.testing.assert_allclose(prog.consts[op.consts["bias"]].host, bt @ k + b, rtol=1e-5, atol=1e-6)
    c1 = prog.consts[op.consts["ln_c1"]].host                    # [N][2][8]: {ca,cb,cc,ca,cb,cc,ca,cb} {cc,0,...}
    terms = pack.bf16_bits_to_f32(c1[:, 0, :3]).astype(np.float64).sum(1)
    np.testing.assert_allclose(terms, pack.bf16_bits_to_f32(wt[:, :192]).astype(np.float64).sum(1), rtol=2e-7, atol=1e-7)
    assert (c1[:, 0, 3:6] == c1[:, 0, :3]).all() and (c1[:, 1, 0] == c1[:, 0, 2]).all() and not c1[:, 1, 1:].any()
    monkeypatch.setenv("TFIMM_NO_LN_FOLD", "1")
    kinds2, _ = _kinds("vit_tiny_patch16_224")
    assert "row_stats" not in kinds2 and kinds2.count("layernorm") == 25
    # the fold exists in the persistent LDS-DMA GEMM family only: switching that family off keeps the LayerNorm launches too
    monkeypatch.delenv("TFIMM_NO_LN_FOLD")
    for switch in ("TFIMM_GEMM_NO_STREAM", "TFIMM_GEMM_NO_DMA"):
        monkeypatch.setenv(switch, "1")
        kinds3, _ = _kinds("vit_tiny_patch16_224")
        assert "row_stats" not in kinds3 and kinds3.count("layernorm") == 25, switch
        monkeypatch.delenv(switch)


def test_constants_reach_a_second_device_after_the_host_copies_were_dropped():
    """Program.upload drops the packed host arrays after the first device upload; a later plan on another device takes the
    constants from the uploaded copy (devices "cpu" then "meta" here: no GPU needed)"""
    m = tfimm.create_model("resnet50_mini_test_model")
    m.set_weights(synthetic_weights(m))
    p = m.program(64, 64)
    p.upload("cpu")
    first = list(p._dev_consts)
    for c in p.consts:
        c.host = None                                    # what an upload to a GPU leaves behind
    p.upload("meta")
    assert len(p._dev_consts) == len(first)
    assert all(str(t.device) == "meta" and t.shape == f.shape and t.dtype == f.dtype for t, f in zip(p._dev_consts, first))


def test_three_way_bf16_split_is_exact_to_24_bits():
    import numpy as np
    from tfimm.engine import pack
    v = np.random.default_rng(0).standard_normal(1000).astype(np.float32) * np.float32(37.0)
    t = pack.bf16_bits_to_f32(pack.split3_bf16(v)).astype(np.float64)
    assert np.abs(t.sum(-1) - v.astype(np.float64)).max() <= np.abs(v).max() * 2.0 ** -23


def test_narrow_mlp_blocks_are_one_launch(monkeypatch):
    """Swin-B / ConvNeXt-B stage 1 (C = 128, hidden = 512): norm -> fc1 -> GELU -> fc2 -> + shortcut is one mlp_fused op; wider
    stages keep the two GEMMs; the switch and the fp32 path lower the plain layers; the flop count does not change."""
    kinds, prog = _kinds("swin_base_patch4_window7_224")
    fused = [op for op in prog.ops if op.kind == "mlp_fused"]
    assert len(fused) == 2 and all(op.attrs["rows"] == 3136 and op.inputs[0] == op.inputs[1] for op in fused)
    flops = prog.flops_per_image()
    kinds, prog = _kinds("convnext_base")
    fused = [op for op in prog.ops if op.kind == "mlp_fused"]
    assert len(fused) == 3 and all(op.inputs[0] != op.inputs[1] for op in fused)      # the shortcut is the block input
    kinds, _ = _kinds("swin_tiny_patch4_window7_224")                                 # C = 96
    assert "mlp_fused" not in kinds
    monkeypatch.setenv("TFIMM_NO_MLP_FUSION", "1")
    kinds, prog = _kinds("swin_base_patch4_window7_224")
    assert "mlp_fused" not in kinds and prog.flops_per_image() == flops


def test_mlp_fused_operands():
    import numpy as np
    from tfimm.engine import pack
    r = np.random.default_rng(5)
    c, h = 128, 512
    k1, k2 = r.standard_normal((c, h)).astype(np.float32), r.standard_normal((h, c)).astype(np.float32)
    b1, b2 = r.standard_normal(h).astype(np.float32), r.standard_normal(c).astype(np.float32)
    gam, bet, ls = (r.standard_normal(c).astype(np.float32) for _ in range(3))
    w1, b1f, w2, b2f = pack.pack_mlp_fused(k1, b1, gam, bet, k2, b2, ls)
    assert w1.shape == (h, c) and w2.shape == (c, h) and w1.dtype == w2.dtype == np.uint16
    np.testing.assert_allclose(pack.bf16_bits_to_f32(w1), (k1 * gam[:, None]).T, rtol=2 ** -8)
    np.testing.assert_allclose(b1f, bet.astype(np.float64) @ k1 + b1, rtol=1e-5, atol=1e-5)
    order = pack.chain_k_order(h)
    assert sorted(order) == list(range(h))
    np.testing.assert_allclose(pack.bf16_bits_to_f32(w2), ((k2 * ls[None, :])[order]).T, rtol=2 ** -8)
    np.testing.assert_allclose(b2f, b2 * ls, rtol=1e-6)


def test_live_tensors_across_a_cut_and_branch_support():
    """Program.live_across(i): what ops [0, i) wrote and ops [i, ...) still need -- the tensors the branches of a hybrid
    recording write into the full plan's buffers.  A ResNet block boundary carries exactly the block output; inside a
    bottleneck the shortcut is live as well.  Programs with the talking-heads launch do not take branches."""
    kinds, prog = _kinds("resnet50")
    ops = prog.ops
    written_before = set()
    for i, op in enumerate(ops):
        if i > 0:
            live = prog.live_across(i)
            assert live and set(live) <= written_before
            needed = {t for o in ops[i:] for t in o.inputs}
            assert all(t in needed or prog.tensors[t].keep for t in live)
        written_before.update(([op.output] if op.output is not None else []) + list(op.extra_outputs))
    # behind the first fused bottleneck tail (stem_pool, conv1, conv_chain): one tensor, 256 channels at 56 x 56
    i = kinds.index("conv_chain") + 1
    (t,) = prog.live_across(i)
    assert (prog.tensors[t].C, prog.tensors[t].rows) == (256, 56 * 56)
    # between conv1 and the tail of the SECOND block: conv1's output and the block input (the shortcut)
    j = [k for k, kd in enumerate(kinds) if kd == "conv_chain"][1]
    assert sorted(prog.tensors[t].C for t in prog.live_across(j)) == [64, 256]
    assert prog.supports_branches()
    _, cait = _kinds("cait_test_model")
    assert cait.supports_branches()          # round 4: the talking-heads kernel is reproducible next to other launches


def _duals(prog):
    return [op for op in prog.ops if op.kind == "gemm" and op.attrs.get("dual")]


def test_strided_shortcut_convolutions_become_a_second_operand_of_conv3(monkeypatch):
    """resnet.py:282-290 + 315-330: the 1x1 / stride-2 `downsample` convolution of the first block of stages 2 - 4 is folded into
    that block's conv3 (tfimm_gemm_desc::a2): three launches and three shortcut tensors fewer, same FLOPs; the stage-1 case
    stays with the chain kernel's own shortcut flavour."""
    kinds, prog = _kinds("resnet50")
    d = _duals(prog)
    assert len(prog.ops) == 46 and len(d) == 3
    assert [(o.attrs["K"], o.attrs["dual"]["K2"], o.attrs["N"], o.attrs["dual"]["stride"]) for o in d] == [
        (128, 256, 512, 2), (256, 512, 1024, 2), (512, 1024, 2048, 2)]
    for o in d:
        du = o.attrs["dual"]
        assert (du["H"], du["W"]) == (2 * du["OH"], 2 * du["OW"]) and o.attrs["act"] == "relu" and not o.attrs.get("has_residual")
        assert o.attrs["ldw"] == -(-o.attrs["K"] // 64) * 64 + -(-du["K2"] // 64) * 64         # both parts padded to whole k-tiles
        assert len(o.inputs) == 2 and prog.tensors[o.inputs[1]].C == du["K2"]
    assert abs(prog.flops_per_image() / 1e9 - 8.178) < 0.01
    monkeypatch.setenv("TFIMM_NO_FOLD_SHORTCUT", "1")
    kinds, plain = _kinds("resnet50")
    assert len(plain.ops) == 49 and not _duals(plain) and abs(plain.flops_per_image() - prog.flops_per_image()) < 1


@pytest.mark.parametrize("name,n", [("resnext50_32x4d", 3), ("wide_resnet50_2", 3), ("resnet101", 3),
                                    ("seresnet50", 0),      # the SE gate sits between conv3 and the add
                                    ("resnet50d", 3),       # average-pool shortcut: four taps of a 2 x 2 window as the second operand
                                    ("seresnet152d", 0),    # ... but not behind an SE gate
                                    ("resnet18", 3),        # basic blocks: the 3 x 3 conv2 gather takes the shortcut as its second operand
                                    ("resnet34", 3),
                                    ("resnet50_gn", 0)])    # GroupNorm does not fold into the weights
def test_which_configurations_fold_their_shortcut(name, n):
    _, prog = _kinds(name)
    assert len(_duals(prog)) == n
