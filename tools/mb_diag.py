"""EfficientNet-B4's fused expand + depthwise launches (batch 256) back to back for a second each, with clock / power telemetry.
    [TFIMM_HIP_LIB=variant.so] python tools/mb_diag.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import hip_ops as H
from tfimm.engine import pack
from telemetry import Telemetry

B = int(os.environ.get("MB_BATCH", "256"))
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(24, 144, 3, 2, 190), (32, 192, 3, 1, 95), (32, 192, 5, 2, 95)]
r = np.random.default_rng(0)
tele = Telemetry(0, hz=250.0, raw=True)
for cin, c, k, s, Hh in shapes:
    x = torch.randn(B, Hh, Hh, cin, device="cuda").to(torch.bfloat16)
    k1 = (r.standard_normal((cin, c)) / np.sqrt(cin)).astype(np.float32)
    kd = (r.standard_normal((k, k, c, 1)) / k).astype(np.float32)
    cpad = pack.ceil_to(c, 32)
    frag = pack.pack_expand_frag(k1, cpad)
    wd, b2 = pack.pack_depthwise(kd, np.ones(c, np.float32), (0.5 * r.standard_normal(c)).astype(np.float32))

    def padc(a):
        out = np.zeros(a.shape[:-1] + (cpad,), np.float32)
        out[..., :c] = a
        return out
    OH = -(-Hh // s)
    tot = max((OH - 1) * s + k - Hh, 0)
    pt = tot // 2
    fd, b1d, wdd, b2d = H.dev_bits(frag), H.dev_f32(padc((0.5 * r.standard_normal(c)).astype(np.float32))), H.dev_f32(padc(wd)), H.dev_f32(padc(b2))
    fn = lambda: H.expand_dwconv(x, fd, b1d, wdd, b2d, c, k, s, pt, pt, OH, OH, act="swish", want_sums=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tele.start(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.0:
        for _ in range(10):
            fn()
        torch.cuda.synchronize(); n += 10
    t1 = time.perf_counter(); tele.stop()
    ss = [q for q in tele.samples if q["t"] > t0 + 0.15]
    sclk = sum(q["sclk"] for q in ss) / max(len(ss), 1); pw = sum(q["power"] for q in ss) / max(len(ss), 1)
    us = (t1 - t0) / n * 1e6
    byts = B * (Hh * Hh * cin + OH * OH * c) * 2
    print(f"expand_dw {cin:3d}->{c:3d} k{k} s{s} {Hh:3d}->{OH:3d}: {us:7.1f} us  {byts / us / 1e6:5.2f} TB/s  sclk {sclk:5.0f} MHz  {pw:5.0f} W", flush=True)
