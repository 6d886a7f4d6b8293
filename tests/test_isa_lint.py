"""The product kernels contain no packed fp32 instruction with a set op_sel bit (tools/isa_lint.py): on gfx950 such an
instruction returns the product of the wrong half in lanes 48..63 while waves of the persistent GEMM kernel share the SIMD --
the round-3 "co-residency defect" (profiles/NOTES_r04.md section 1).  Compiles every translation unit to device assembly
(hipcc cross-compiles without a GPU) and scans it; about a minute on 8 cores."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_no_packed_fp32_operand_select_in_product_kernels(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_lint.py"), "--keep", str(tmp_path)], capture_output=True,
                       text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 packed-fp32 instructions with a set op_sel bit" in r.stdout


def test_the_lint_sees_what_it_is_looking_for(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    f = tmp_path / "k.s"
    f.write_text("_Z1kv:\n\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0,0]\n"
                 "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]\n\tv_pk_add_f32 v[0:1], v[2:3], v[4:5]\n\ts_endpgm\n")
    hits = isa_lint.scan(str(f))
    assert [k for k, _ in hits] == ["_Z1kv", "_Z1kv"] and "op_sel:[0,1,0]" in hits[0][1] and "v_pk_mul_f32" in hits[1][1]
