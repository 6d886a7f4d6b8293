"""pytest -m gpu: every HIP kernel against the CPU oracle ops (cases in hip_checks.py)."""
import pytest

import hip_checks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(hip_checks.CASES))
def test_op(name):
    err, tol = hip_checks.run_case(name)
    assert err <= tol, f"{name}: rel-to-max error {err:.3e} > {tol:.1e}"
