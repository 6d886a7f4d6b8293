"""Run hip_checks cases whose name contains one of the given substrings:  python tools/run_cases.py dwconv expand_dw ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import hip_checks
bad = 0
for n in sorted(hip_checks.CASES):
    if any(k in n for k in sys.argv[1:]):
        t = time.time()
        try:
            e, tol = hip_checks.run_case(n)
            ok = e <= tol
            bad += not ok
            print(f"{'ok  ' if ok else 'FAIL'} {n:52s} {e:10.3e} of {tol:.1e}  ({time.time() - t:.1f} s)", flush=True)
        except Exception as ex:
            bad += 1
            print(f"FAIL {n:52s} {type(ex).__name__}: {ex}", flush=True)
print("failed:", bad)
sys.exit(1 if bad else 0)
