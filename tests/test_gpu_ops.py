"""pytest -m gpu: every HIP kernel against the CPU oracle ops (cases in hip_checks.py)."""
import pytest

import hip_checks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(hip_checks.CASES))
def test_op(name):
    err, tol = hip_checks.run_case(name)
    assert err <= tol, f"{name}: rel-to-max error {err:.3e} > {tol:.1e}"


def test_conv_chain_splits_oversize_batches_into_image_chunks():
    """tfimm_hip_conv_chain runs a batch whose activations exceed the 2 GiB a buffer descriptor addresses as image chunks
    (ResNet-50 stage 1 from batch 1338 on).  TFIMM_CHAIN_LIMIT lowers that threshold (read once per process, hence the
    subprocess): 3 images per chunk here, 14 launches for the 40-image cases -- same oracle, same tolerance."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import hip_checks\n"
            "for n in ('chain_56x56_b3', 'chain_multiround_b40', 'chain_shortcut_conv_multiround_b40', 'chain_n512'):\n"
            "    e, t = hip_checks.run_case(n); print(n, e, t); assert e <= t, (n, e, t)\n"
            % (root, os.path.join(root, "tensorflow-image-models_amd"), os.path.join(root, "tests")))
    env = dict(os.environ, TFIMM_CHAIN_LIMIT=str(5 * 1000 * 1000))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
