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
def test_talking_heads_programs_keep_one_branch():
    """CaiT: the talking-heads launch is not bit-reproducible next to launches of another stream (Program.supports_branches,
    profiles/NOTES_r03.md section 9) -- such programs ignore ``branches`` and stay reproducible."""
    import model_checks as mc
    model = tfimm.create_model("cait_xxs24_224")
    model.set_weights(synthetic_weights(model, 2021))
    assert not model.program().supports_branches()
    x = mc.make_input(model.cfg, 8)
    want = model(x).numpy()
    model.branches = 2
    runs = [model(x).numpy() for _ in range(3)]
    assert all(np.array_equal(r, want) for r in runs)
    assert not [k for k in model._plans if "branches" in k]


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
