"""Compare gpurun_out/features_<name>.npz (tools/feature_forward.py, GPU) with the fp32 oracle, feature by feature, next to
the oracle's own sensitivity to bf16-rounded kernels + input (the engine stores both in bf16).
    python tools/feature_check.py <name> [...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("TFIMM_ALLOW_NO_GPU", "1")
import numpy as np
import torch
import tfimm, oracle
from tfimm.utils.init import synthetic_weights
import model_checks as mc


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


for name in sys.argv[1:]:
    got = np.load(os.path.join(ROOT, "gpurun_out", f"features_{name}.npz"))
    m = tfimm.create_model(name)
    w = synthetic_weights(m, 2021)
    x = mc.make_input(m.cfg, 2, 2021)
    ref, rf = oracle.forward(m.cfg, w, x, return_features=True)
    w2 = {k: (bf(v) if k.split(":")[0].endswith("kernel") and v.ndim >= 2 else v) for k, v in w.items()}
    r2, rf2 = oracle.forward(m.cfg, w2, bf(np.asarray(x, dtype=np.float32)), return_features=True)
    print(f"# {name}: rel-to-max error of the engine against the fp32 oracle | of the oracle with bf16-rounded kernels + input against itself")
    for k in rf:
        if "f:" + k not in got.files:
            continue
        a = np.asarray(rf[k], dtype=np.float32).reshape(-1)
        st = max(1, a.size // 16384)
        a, b2 = a[::st][:16384], np.asarray(rf2[k], dtype=np.float32).reshape(-1)[::st][:16384]
        print(f"{k:28s} {mc.rel_err(got['f:' + k], a):.3e} | {mc.rel_err(b2, a):.3e}")
    print(f"{'logits':28s} {mc.rel_err(got['logits'].reshape(np.asarray(ref).shape), ref):.3e} | {mc.rel_err(r2, ref):.3e}")
