"""Depthwise launches of EfficientNet-B4 (batch 256) back to back for a second each, with clock / power telemetry.
    python tools/dw_diag.py [C:H:k:s ...]      (default: the B4 shapes that run on dwconv_rows_kernel)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import hip_ops as H
from telemetry import Telemetry

shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(336, 48, 5, 1), (672, 24, 3, 1), (960, 24, 5, 1), (1632, 12, 5, 1), (2688, 12, 3, 1)]
B = 256
tele = Telemetry(0, hz=250.0, raw=True)
for C, Hh, k, s in shapes:
    x = torch.randn(B, Hh, Hh, C, device="cuda").to(torch.bfloat16)
    w = torch.randn(k * k, C, device="cuda") * 0.2
    b = torch.randn(C, device="cuda")
    pad = k // 2
    OH = (Hh + 2 * pad - k) // s + 1
    fn = lambda: H.dwconv(x, w, b, k, s, pad, pad, OH, OH, act="swish", want_sums=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tele.start(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.0:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    t1 = time.perf_counter(); tele.stop()
    ss = [q for q in tele.samples if q["t"] > t0 + 0.15]
    sclk = sum(q["sclk"] for q in ss) / max(len(ss), 1); pw = sum(q["power"] for q in ss) / max(len(ss), 1)
    us = (t1 - t0) / n * 1e6
    byts = B * (Hh * Hh + OH * OH) * C * 2
    print(f"dwconv C={C:5d} {Hh:3d}x{Hh:<3d} k{k} s{s}: {us:7.1f} us  {byts / us / 1e6:5.2f} TB/s  {2.0 * B * OH * OH * C * k * k / us / 1e6:5.1f} TF/s  sclk {sclk:5.0f} MHz  {pw:5.0f} W", flush=True)
