"""Registers and spills of every product kernel, this tree against a git revision (CPU only: hipcc cross-compiles).

    python tools/kernel_regs_diff.py [REV]        # default HEAD~1
Compiles every translation unit of both trees to device assembly (tools/isa_lint.py's compile_unit) and prints the kernels whose
.vgpr_count moved by >= 8, whose .vgpr_spill_count changed, or whose .sgpr_spill_count grew by > 4 -- the check that would
have caught round 4's expand_dw regression (profiles/NOTES_r04.md section 7) before a GPU run."""
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402


def build(srcdir, outdir):
    os.makedirs(outdir, exist_ok=True)
    isa_lint.SRC = srcdir
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        return list(ex.map(isa_lint.compile_unit, [(u, outdir, []) for u in isa_lint.units()]))


def meta(files):
    out = {}
    for f in files:
        s = open(f).read()
        md = s[s.find("amdhsa.kernels:"):]
        for blk in md.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s*(\S+)", blk).group(1)
            get = lambda k: int(re.search(re.escape(k) + r":\s*(\d+)", blk).group(1))  # noqa: E731
            out[name] = (get(".vgpr_count"), get(".vgpr_spill_count"), get(".sgpr_spill_count"))
    return out


def main():
    rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD~1"
    tmp = tempfile.mkdtemp(prefix="regs_diff_")
    subprocess.run(f"git -C {ROOT} archive {rev} tensorflow-image-models_amd/csrc include | tar x -C {tmp}", shell=True, check=True)
    old = meta(build(os.path.join(tmp, "tensorflow-image-models_amd", "csrc"), os.path.join(tmp, "old_s")))
    new = meta(build(os.path.join(ROOT, "tensorflow-image-models_amd", "csrc"), os.path.join(tmp, "new_s")))
    n = 0
    for k in sorted(new):
        if k in old and (new[k][1] != old[k][1] or abs(new[k][0] - old[k][0]) >= 8 or new[k][2] > old[k][2] + 4):
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            name = name.replace("tfimm_gemm::", "").replace("(anonymous namespace)::", "")[:100]
            print(f"{name:100s} (vgpr, vgpr spills, sgpr spills)  {rev} {old[k]}  ->  tree {new[k]}")
            n += 1
    print(f"{n} kernels changed notably; {len(new)} kernels in the tree, {len(set(new) - set(old))} new, {len(set(old) - set(new))} gone")


if __name__ == "__main__":
    main()
