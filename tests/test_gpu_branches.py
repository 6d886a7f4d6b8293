"""A batch as slices on parallel branches of one HIP graph (engine/graph.py CapturedBranches, Model.branches): the same
kernels on the same operands, so the result must equal the single-plan forward BIT FOR BIT -- eagerly (first call) and
from the recorded graph (later calls), for even and uneven slices."""
import numpy as np
import pytest

import test_architectures  # noqa: F401
import tfimm
from tfimm.engine.graph import branch_sizes
from tfimm.utils.init import synthetic_weights


def test_branch_sizes():
    assert branch_sizes(256, 2) == [128, 128]
    assert branch_sizes(7, 2) == [4, 3] and branch_sizes(7, 3) == [3, 2, 2]
    assert branch_sizes(2, 5) == [1, 1] and branch_sizes(5, 1) == [5] and branch_sizes(5, 0) == [5]


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch,parts", [("resnet50", 8, 2), ("swin_tiny_patch4_window7_224", 7, 2),
                                              ("efficientnet_b0", 9, 3), ("convnext_test_model", 6, 2)])
def test_parallel_branches_reproduce_the_single_plan_bit_for_bit(name, batch, parts):
    import model_checks as mc
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = mc.make_input(model.cfg, batch)
    want = model(x).numpy()
    model.branches = parts
    first = model(x).numpy()            # slices launched one after the other
    second = model(x).numpy()           # one hipGraphLaunch, parallel branches
    third = model(x).numpy()
    assert np.array_equal(first, want) and np.array_equal(second, want) and np.array_equal(third, want)
    y, feats = model(x, return_features=True)
    model.branches = 1
    y1, feats1 = model(x, return_features=True)
    assert np.array_equal(y.numpy(), y1.numpy())
    assert all(np.array_equal(feats[k].numpy(), feats1[k].numpy()) for k in feats1)


@pytest.mark.gpu
def test_small_batches_keep_one_branch():
    import model_checks as mc
    model = tfimm.create_model("resnet_test_model_1")
    model.set_weights(synthetic_weights(model, 2021))
    model.branches = 2
    x = mc.make_input(model.cfg, 3)
    model(x)
    assert not [k for k in model._plans if "branches" in k]       # fewer than 2 images per branch: the plain path


@pytest.mark.gpu
def test_talking_heads_programs_run_as_branches_bit_for_bit():
    """CaiT under parallel branches (excluded in round 3: profiles/NOTES_r04.md section 1 has the cause and the fix): every
    replay of the forked recording gives the single-plan bits."""
    import model_checks as mc
    model = tfimm.create_model("cait_xxs24_224")
    model.set_weights(synthetic_weights(model, 2021))
    assert model.program().supports_branches()
    x = mc.make_input(model.cfg, 64)
    want = model(x).numpy()
    model.branches = 2
    runs = [model(x).numpy() for _ in range(8)]
    assert [k for k in model._plans if "branches" in k]
    assert all(np.array_equal(r, want) for r in runs)


@pytest.mark.gpu
def test_talking_heads_next_to_the_gemm_that_disturbed_it():
    """The pair of launches that was not reproducible in round 3, 60 times on two streams: talking-heads attention (heads 4,
    hd 48, 196 tokens) while 768 -> 192 GEMMs with residual on the 256 x 64 tiles (four and eight waves) run back to back on
    another stream.  Every talking-heads result must equal the solo run bit for bit, and so must the GEMM's."""
    import math

    import torch

    import hip_ops as H
    from tfimm.engine import pack
    B, N, heads, hd = 64, 196, 4, 48
    r = np.random.default_rng(1)
    g = torch.Generator(device="cuda").manual_seed(2)
    qkv = torch.randn(B * N, 3 * heads * hd, device="cuda", generator=g).to(torch.bfloat16)
    wl = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
    ww = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
    bl = (0.3 * r.standard_normal(heads)).astype(np.float32)
    bw = (0.02 * r.standard_normal(heads)).astype(np.float32)

    def tha():
        return H.talking_heads_attention(qkv, B, N, heads, hd, hd ** -0.5, wl, bl, ww, bw)

    ref = tha().view(torch.int16).clone()
    torch.cuda.synchronize()
    M, K, Nn = 12544, 768, 192
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    wt, _ = pack.pack_dense((r.standard_normal((K, Nn)) / math.sqrt(K)).astype(np.float32), None)
    wd, bias = H.dev_bits(wt), H.dev_f32(r.standard_normal(Nn).astype(np.float32))
    res = torch.randn(M, Nn, device="cuda", generator=g).to(torch.bfloat16)
    outs = {hint: torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda") for hint in (27, 24)}

    def gemm(hint):
        return H.gemm(a, wd, Nn, K, bias=bias, residual=res, tile_hint=hint, out=outs[hint])

    gref = {}
    for hint in outs:
        gemm(hint)
        torch.cuda.synchronize()
        gref[hint] = outs[hint].view(torch.int16).clone()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = gbad = 0
    for rep in range(60):
        hint = (27, 24)[rep & 1]
        with torch.cuda.stream(sb):
            for _ in range(6):
                gemm(hint)
        with torch.cuda.stream(sa):
            o = tha()
        torch.cuda.synchronize()
        bad += int(not torch.equal(o.view(torch.int16), ref))
        gbad += int(not torch.equal(outs[hint].view(torch.int16), gref[hint]))
    assert bad == 0 and gbad == 0, f"talking-heads results differing from the solo run: {bad} / 60; GEMM results: {gbad} / 60"


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("resnet50", 8), ("swin_tiny_patch4_window7_224", 6), ("efficientnet_b0", 7),
                                        ("convnext_test_model", 6), ("seresnet_test_model", 5)])
def test_hybrid_recordings_reproduce_the_single_plan_bit_for_bit(name, batch):
    """CapturedHybrid: two half-batch branches for ops [0, cut), the full-batch plan for the rest, what crosses the join
    written by the branches straight into the full plan's buffers -- every cut position tried must give the single plan's bits."""
    import torch

    import model_checks as mc
    from tfimm.engine.graph import CapturedHybrid
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = torch.from_numpy(mc.make_input(model.cfg, batch)).cuda().contiguous()
    prog = model.program()
    plan = prog.make_plan(batch)
    plan.run(x)
    torch.cuda.synchronize()
    out_t = prog.outputs["logits"]
    want = plan.tensor_view(out_t).float().cpu().numpy().copy()
    n = len(prog.ops)
    for cut in sorted({1, 2, n // 3, n // 2, (2 * n) // 3, n - 1, n}):
        h = CapturedHybrid(prog, x, cut)
        for _ in range(2):
            h.replay()
        torch.cuda.synchronize()
        got = h.output(out_t).float().cpu().numpy()
        assert np.array_equal(got, want), (name, cut, float(np.abs(got - want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("resnet50", 8), ("vit_tiny_patch16_224", 6)])
def test_recorded_sink_delivers_the_logits_with_the_replay(name, batch):
    """``sink=(output, tensor)``: the recording's last node copies the program output into the caller's tensor, for the plain
    recording, parallel branches and the hybrid -- the step bench.py times is then one hipGraphLaunch.  Every replay must leave
    the bits of the eager forward there."""
    import torch

    import model_checks as mc
    from tfimm.engine.graph import CapturedBranches, CapturedHybrid
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = torch.from_numpy(mc.make_input(model.cfg, batch)).cuda().to(torch.bfloat16)
    prog = model.program()
    out_t = prog.outputs["logits"]
    plan = prog.make_plan(batch)
    plan.run(x)
    want = plan.tensor_view(out_t).view(batch, out_t.C).float().clone()
    recs = []
    for kind in ("plain", "branches", "hybrid"):
        dst = torch.full((batch, out_t.C), float("nan"), dtype=torch.float32, device="cuda")
        if kind == "plain":
            rec = prog.make_plan(batch).capture(x, sink=(out_t, dst))
        elif kind == "branches":
            rec = CapturedBranches(prog.make_branches(batch, 2), x, sink=(out_t, dst))
        else:
            rec = CapturedHybrid(prog, x, max(1, len(prog.ops) // 2), sink=(out_t, dst))
        recs.append((kind, rec, dst))
    for kind, rec, dst in recs:
        for _ in range(3):
            dst.fill_(float("nan"))
            rec.replay()
            torch.cuda.synchronize()
            assert torch.equal(dst, want), kind
