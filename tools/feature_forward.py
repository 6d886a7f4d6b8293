"""Forward one configuration on the GPU with return_features=True and save logits + every feature map
(a strided sample of 16 k values each, gpurun_out/features_<name>.npz) -- tools/feature_check.py compares them with the fp32 oracle on a CPU box, feature by
feature: where along the depth of a network does the deviation from the oracle grow?
    python tools/feature_forward.py <name> [...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import tfimm
from tfimm.utils.init import synthetic_weights
import model_checks as mc

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name in sys.argv[1:]:
    m = tfimm.create_model(name)
    m.set_weights(synthetic_weights(m, 2021))
    x = mc.make_input(m.cfg, 2, 2021)
    y, feats = m(x, return_features=True)
    out = {"logits": y.numpy().astype(np.float32)}
    for k, v in feats.items():
        a = np.asarray(v.numpy() if hasattr(v, "numpy") else v, dtype=np.float32).reshape(-1)
        out["f:" + k] = a[::max(1, a.size // 16384)][:16384]          # a strided sample (the merge-back limit is 64 MiB)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"features_{name}.npz"), **out)
    print(name, len(feats), "features saved")
