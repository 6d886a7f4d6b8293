"""Measure the best GEMM tile for every GEMM of the benchmark workloads on this GPU and write
tfimm/engine/gemm_tune.json (copy under gpurun_out/ too so it travels back from a GPU box).

  python tools/tune_gemm.py [model:batch ...]
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402,F401

DEFAULT = ["resnet50:256", "vit_base_patch16_224:512", "swin_base_patch4_window7_224:256", "efficientnet_b4:256",
           "convnext_tiny:256", "cait_xxs24_224:256", "vit_tiny_patch16_224:1", "vit_tiny_patch16_224:2", "resnet50:8",
           "resnet50:2"]


def main():
    import tfimm
    from tfimm.engine import tune
    from tfimm.utils.init import synthetic_weights
    specs = sys.argv[1:] or DEFAULT
    verbose = os.environ.get("TUNE_VERBOSE", "0") == "1"
    for spec in specs:
        name, b = spec.split(":")
        m = tfimm.create_model(name)
        m.set_weights(synthetic_weights(m))
        plan = m.program().make_plan(int(b))
        n = plan.autotune(iters=5, verbose=verbose)
        cfg = m.cfg
        x = torch.randn(int(b), *cfg.input_size, cfg.in_channels, device="cuda").to(torch.bfloat16)
        ch = plan.autotune_in_context(x, top=4, iters=3, verbose=verbose) if n else 0
        print(f"{spec}: tuned {n} new shapes ({ch} changed by the in-context pass), table size {len(tune.TABLE)}", flush=True)
        del plan, m
        torch.cuda.empty_cache()
    tune.save()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(ROOT, "tensorflow-image-models_amd", "tfimm", "engine", "gemm_tune.json"),
                os.path.join(out, "gemm_tune.json"))
    from collections import Counter
    print("hint histogram:", sorted(Counter(tune.TABLE.values()).items()))


if __name__ == "__main__":
    main()
