"""The fp32 oracle's OWN sensitivity to bf16 storage of the kernels and the input, for the configurations whose depth
amplifies rounding beyond the general 5e-2 bar (EfficientNetV2-XL: 8 + 8 + 16 + 24 + 32 + 8 blocks behind a 4-block stage
0; random-init weights):  s = max |oracle(bf16(W), bf16(x)) - oracle(W, x)| / max |oracle(W, x)|  on the test input
(model_checks.make_input, batch 1, seed 2021).  No engine involved: it is a property of the model + weights, computed on the
CPU here, and it is what a bf16 forward of that model can be held to -- an engine that stores activations in bf16 as well
cannot be expected below it, and must not be far above it.

Writes the `deep_configs` section of tests/golden/bf16_bars.json:  logits_bar = s rounded up to two digits (the engine's
observed error on an MI355X is recorded next to it by hand from the GPU run; tests/test_gpu_models.py enforces
engine <= logits_bar for the bf16 path and <= 1e-3 for the float32 path of the same program).

    python tools/make_bf16_sensitivity.py [name ...]
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("TFIMM_ALLOW_NO_GPU", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import model_checks as mc  # noqa: E402
import oracle  # noqa: E402
import tfimm  # noqa: E402
from tfimm.utils.init import synthetic_weights  # noqa: E402

DEFAULT = ["efficientnet_v2_xl_in21k", "efficientnet_v2_xl_in21ft1k"]
BARS = os.path.join(ROOT, "tests", "golden", "bf16_bars.json")


def bf(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def ceil2(v):
    e = math.floor(math.log10(v)) - 1
    return round(math.ceil(v / 10 ** e) * 10 ** e, 6)


def main():
    names = sys.argv[1:] or DEFAULT
    with open(BARS) as f:
        bars = json.load(f)
    deep = bars.setdefault("deep_configs", {})
    deep["_doc"] = ("configurations held to a per-config bar instead of the general 5e-2: logits_bar = the fp32 oracle's own "
                    "rel-to-max change when its conv / dense kernels and its input are rounded to bf16 (tools/make_bf16_sensitivity.py, "
                    "batch 1, seed 2021), rounded up to two digits; engine_observed = the bf16 engine on an MI355X (bit-reproducible)")
    for name in names:
        m = tfimm.create_model(name)
        w = synthetic_weights(m, 2021)
        x = mc.make_input(m.cfg, 1, 2021)
        ref = np.asarray(oracle.forward(m.cfg, w, x))
        w2 = {k: (bf(v) if k.split(":")[0].endswith("kernel") and v.ndim >= 2 else v) for k, v in w.items()}
        r2 = np.asarray(oracle.forward(m.cfg, w2, bf(np.asarray(x, dtype=np.float32))))
        s = float(mc.rel_err(r2, ref))
        entry = deep.setdefault(name, {})
        entry.update(oracle_bf16_sensitivity=round(s, 5), logits_bar=ceil2(s), batch=1, input_size=list(m.cfg.input_size))
        print(name, entry, flush=True)
    with open(BARS, "w") as f:
        json.dump(bars, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
