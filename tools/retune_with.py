"""Let new GEMM tile hints challenge the committed table IN CONTEXT: for every GEMM shape of the given workloads the table's
current choice and the challengers are timed inside full forwards (HIP events around the launches of that shape), and a
challenger replaces the incumbent when it is >= MIN_GAIN faster.  Cheap (a few forwards per shape) -- unlike a re-tune from scratch.

  python tools/retune_with.py 30 [model:batch ...]        -> tfimm/engine/gemm_tune.json (+ copy under gpurun_out/)
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

DEFAULT = ["resnet50:256", "vit_base_patch16_224:512", "swin_base_patch4_window7_224:256", "efficientnet_b4:256",
           "convnext_tiny:256", "cait_xxs24_224:256"]
MIN_GAIN = float(os.environ.get("RETUNE_MIN_GAIN", "0.02"))


def main():
    import tfimm
    from tfimm.engine import tune
    from tfimm.utils.init import synthetic_weights
    challengers = tuple(int(v) for v in sys.argv[1].split(","))
    for spec in sys.argv[2:] or DEFAULT:
        name, b = spec.split(":")
        m = tfimm.create_model(name)
        m.set_weights(synthetic_weights(m))
        plan = m.program().make_plan(int(b))
        cfg = m.cfg
        x = torch.randn(int(b), *cfg.input_size, cfg.in_channels, device="cuda").to(torch.bfloat16)
        before = {tune.key_of(d): tune.TABLE.get(tune.key_of(d), 0) for d in plan._gemm_descs}
        ch = plan.autotune_in_context(x, iters=3, verbose=os.environ.get("TUNE_VERBOSE") == "1", challengers=challengers,
                                      min_gain=MIN_GAIN)
        moved = [(k, before[k], tune.TABLE[k]) for k in before if tune.TABLE.get(k) != before[k]]
        print(f"{spec}: {len(before)} shapes, {ch} changed: {moved}", flush=True)
        del plan, m
        torch.cuda.empty_cache()
    tune.save()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(ROOT, "tensorflow-image-models_amd", "tfimm", "engine", "gemm_tune.json"), os.path.join(out, "gemm_tune.json"))


if __name__ == "__main__":
    main()
