"""pytest -m gpu: the float32 VERIFICATION PATH of the engine (TFIMM_PRECISION=fp32, tfimm/engine/precision.py,
csrc/ref32.hip) against the outputs of THE REFERENCE'S OWN MODEL CODE (tests/golden/forward_golden.npz) at the
reference's own bar: 1e-3 relative to the maximum (tests/test_timm.py:71 of the reference).

Same lowering, same host-side weight transformations (folded BatchNorm, LayerScale, SE gate as an operand scale, remapped
token rows, window index maps ...) as the bf16 product path, minus the cross-layer fusions; float32 storage and
arithmetic.  What passes here is therefore a statement about the engine's SEMANTICS -- epsilons, exact-erf GELU, padding,
pooling conventions, roll / window / mask arithmetic -- that the bf16 bars (tests/golden/bf16_bars.json) are too wide to make.
"""
import os

import numpy as np
import pytest

import model_checks as mc
import test_architectures  # noqa: F401
import tfimm
from tfimm.engine import precision
from tfimm.utils.init import synthetic_weights

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "forward_golden.npz"))
MODELS = sorted({k.split("/")[0] for k in GOLD.files})
TOL_FP32 = 1e-3          # the reference's fp32-vs-fp32 bar (tests/test_timm.py:71)


def _features(name):
    pre = f"{name}/feat/"
    return [k[len(pre):] for k in GOLD.files if k.startswith(pre)]


@pytest.mark.parametrize("name", MODELS)
def test_fp32_engine_matches_reference_code_path_at_1e3(name):
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    ref = GOLD[f"{name}/logits"]
    x = mc.make_input(model.cfg, ref.shape[0])
    with precision.use("fp32"):
        frozen = _features(name)
        if frozen:
            got, feats = model(x, return_features=True)
            assert list(feats.keys()) == frozen
        else:
            got, feats = model(x), {}
        assert got.torch().dtype.is_floating_point and got.torch().element_size() == 4
        err = mc.rel_err(got.numpy().reshape(ref.shape), ref)
        assert err <= TOL_FP32, f"{name}: logits rel-to-max {err:.2e}"
        assert (got.numpy().reshape(ref.shape).argmax(-1) == ref.argmax(-1)).all()
        for k in frozen:
            r = GOLD[f"{name}/feat/{k}"]
            e = mc.rel_err(feats[k].numpy().reshape(r.shape), r)
            assert e <= TOL_FP32, f"{name}: feature {k} rel-to-max {e:.2e}"


def test_fp32_and_bf16_plans_coexist_and_fp32_accepts_uint8():
    """one model object serves both precisions (separate programs / plans), and the fp32 path takes the deferred uint8
    preprocessing like the bf16 path (models/factory.py:165-167 in float32)"""
    name = "resnet50_mini_test_model"
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (2, *model.cfg.input_size, 3), dtype=np.uint8)
    pre = tfimm.create_preprocessing(name, dtype="float32")
    pre_dev = tfimm.create_preprocessing(name, defer=True)
    y_bf16 = model(pre(img)).numpy()
    with precision.use("fp32"):
        y32 = model(pre(img)).numpy()
        y32_u8 = model(pre_dev(img)).numpy()
    y_bf16_again = model(pre(img)).numpy()
    assert np.array_equal(y_bf16, y_bf16_again)
    assert mc.rel_err(y32_u8, y32) <= 1e-5
    assert 0 < mc.rel_err(y_bf16, y32) <= mc.TOL_LOGITS
