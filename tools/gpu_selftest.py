"""Run every op/model parity case on the GPU without stopping at failures and write a table
to gpurun_out/selftest.log; then a quick GEMM / model throughput probe.

Usage (on a GPU box):  python tools/gpu_selftest.py [--skip-perf]
"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "selftest.log"), "w")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()


def run_ops():
    import hip_checks
    nfail = 0
    for name in sorted(hip_checks.CASES):
        try:
            err, tol = hip_checks.run_case(name)
            ok = err <= tol
            log(f"{'PASS' if ok else 'FAIL'} op {name:48s} err={err:.3e} tol={tol:.1e}")
            nfail += 0 if ok else 1
        except Exception as e:  # noqa: BLE001
            nfail += 1
            log(f"EXC  op {name:48s} {type(e).__name__}: {e}")
            log(traceback.format_exc(limit=3))
            try:
                torch.cuda.synchronize()
            except Exception as e2:  # noqa: BLE001
                log("device unusable after exception:", e2)
                return nfail
    return nfail


def run_models(names):
    import model_checks as mc
    import test_architectures  # noqa: F401
    nfail = 0
    for name, batch, feats in names:
        try:
            t = time.time()
            r = mc.compare_model(name, batch=batch, features=feats)
            ok = r["logits"] <= mc.TOL_LOGITS
            worst = {k: f"{v:.2e}" for k, v in r.items() if k.startswith("feat:")}
            log(f"{'PASS' if ok else 'FAIL'} model {name:36s} B={batch} logits_err={r['logits']:.3e} "
                f"top1={r['top1_agree']:.2f} shape={r['shape']} t={time.time() - t:.1f}s {worst if feats else ''}")
            nfail += 0 if ok else 1
        except Exception as e:  # noqa: BLE001
            nfail += 1
            log(f"EXC  model {name} {type(e).__name__}: {e}")
            log(traceback.format_exc(limit=6))
    return nfail


def gemm_perf():
    import hip_ops as H
    from tfimm.engine import pack
    shapes = [(100864, 768, 2304), (100864, 768, 768), (100864, 768, 3072), (100864, 3072, 768),
              (8192, 8192, 8192), (802816, 64, 256), (802816, 256, 64), (200704, 512, 128), (12544, 2048, 512)]
    for (M, K, N) in shapes:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for tile in (1, 4, 11, 12, 13, 14, 15, 16):
            try:
                for _ in range(2):
                    H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=tile)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = 5
                e0.record()
                for _ in range(iters):
                    H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=tile)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                tf = 2.0 * M * N * K / ms / 1e9
                gb = (M * K + N * K + M * N) * 2 / ms / 1e6
                log(f"perf gemm M={M} K={K} N={N} tile={tile} {ms:.3f} ms {tf:.1f} TFLOP/s {gb:.0f} GB/s")
            except Exception as e:  # noqa: BLE001
                log(f"perf gemm M={M} K={K} N={N} tile={tile} EXC {e}")
        del a, w, out


def model_perf():
    import tfimm
    from tfimm.utils.init import synthetic_weights
    for name, B, mbs in [("vit_base_patch16_224", 512, (None, 128)), ("resnet50", 256, (None, 64, 32)),
                         ("vit_tiny_patch16_224", 512, (None,)),
                         ("swin_base_patch4_window7_224", 256, (None,)), ("efficientnet_b4", 256, (None, 64))]:
        try:
            m = tfimm.create_model(name)
            m.set_weights(synthetic_weights(m))
            x = torch.randn(B, *m.cfg.input_size, 3, device="cuda").to(torch.bfloat16)
            prog = m.program()
            for mb in mbs:
                nb = mb or B
                plan = prog.make_plan(nb)
                def step():
                    for s in range(0, B, nb):
                        plan.run(x[s:s + nb])
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = 5
                e0.record()
                for _ in range(iters):
                    step()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / iters
                fl = prog.flops_per_image() * B
                log(f"perf model {name} B={B} micro_batch={nb} {ms:.2f} ms/step {B / ms * 1e3:.0f} img/s "
                    f"{fl / ms / 1e9:.1f} TFLOP/s")
                del plan
            del m, x
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            log(f"perf model {name} EXC {type(e).__name__}: {e}")
            log(traceback.format_exc(limit=6))


if __name__ == "__main__":
    log("device:", torch.cuda.get_device_name(0), "torch", torch.__version__)
    f1 = run_ops()
    f2 = run_models([("vit_test_model", 3, True), ("deit_test_model", 3, False), ("vit_hd64_test_model", 3, False),
                     ("resnet_test_model_1", 3, True), ("resnet_test_model_2", 3, False),
                     ("resnet50_mini_test_model", 3, False), ("seresnet_test_model", 3, False),
                     ("vit_tiny_patch16_224", 2, False), ("resnet18", 2, False), ("resnet50", 2, False),
                     ("vit_base_patch16_224", 1, False), ("swin_test_model", 3, True), ("swin_shift_test_model", 3, True),
                     ("efficientnet_test_model", 3, True), ("efficientnet_same_test_model", 3, True),
                     ("swin_tiny_patch4_window7_224", 2, False), ("efficientnet_b0", 2, False),
                     ("efficientnet_v2_b0", 2, False), ("mobilenet_v2_100", 2, False), ("efficientnet_es", 2, False),
                     ("swin_base_patch4_window7_224", 1, False), ("efficientnet_b4", 1, False)])
    log(f"SUMMARY op_failures={f1} model_failures={f2}")
    if "--skip-perf" not in sys.argv:
        gemm_perf()
        model_perf()
    log("DONE")
