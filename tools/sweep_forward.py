"""Forward every registered configuration (below a parameter budget) once on the GPU with the synthetic weights and save
the logits -- tools/sweep_check.py compares them with the fp32 oracle on a CPU box.
    python tools/sweep_forward.py [max_params_millions] [name_filter] [min_params_millions]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import tfimm
from tfimm.utils.init import synthetic_weights
import model_checks as mc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
flt = sys.argv[2] if len(sys.argv) > 2 else ""
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
from tfimm.engine import precision
# TFIMM_PRECISION=fp32: the float32 verification path (csrc/ref32.hip) -> gpurun_out/sweep_fp32, checked at 1e-3
out = os.path.join(ROOT, "gpurun_out", "sweep_fp32" if precision.get() == "fp32" else "sweep")
os.makedirs(out, exist_ok=True)
names = [n for n in tfimm.list_models() if flt in n]
t0 = time.time()
done = skipped = 0
for name in names:
    try:
        m = tfimm.create_model(name)
        nparams = sum(int(np.prod(s.shape)) for s in m.weight_specs().values())
        if nparams > budget * 1e6 or nparams <= floor * 1e6:
            skipped += 1
            continue
        m.set_weights(synthetic_weights(m, 2021))
        x = mc.make_input(m.cfg, 2, 2021)
        y = m(x).numpy()
        np.save(os.path.join(out, name + ".npy"), y.astype(np.float32))
        done += 1
    except Exception as e:  # noqa: BLE001
        with open(os.path.join(out, name + ".err"), "w") as f:
            f.write(f"{type(e).__name__}: {e}\n")
    del m
print(f"{done} models forwarded, {skipped} above {budget:.0f} M parameters, {time.time() - t0:.0f} s")
