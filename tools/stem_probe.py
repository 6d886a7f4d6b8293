"""NEEDS A PROBE BUILD: the *_DBG switches / stamps exist only with -DTFIMM_PROBE_HOOKS (tools/probes/build_dbg_libs.sh all; TFIMM_HIP_LIB=tools/probes/bin/libtfimm_hip_probe.so).
Time the ResNet stem (7x7 s2 RGB conv on the pixel-pair view) per tile hint, optional cycle stamps of workgroup 0:
   stem_probe.py [B] [hints comma-separated] [stamps 0/1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hints = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,21,22,23,24,25,26,27,29").split(",")]
stamps = len(sys.argv) > 3 and sys.argv[3] == "1"
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
if stamps:
    os.environ["TFIMM_GEMM_DBG"] = "64"
    os.environ["TFIMM_GEMM_DBG_PTR"] = hex(buf.data_ptr())
import hip_ops as H
from tfimm.engine import pack
k, stride, Cout, Hh, Ww = 7, 2, 64, 224, 224
r = np.random.default_rng(0)
kern = (r.standard_normal((k, k, 3, Cout)) / 12).astype(np.float32)
wt, bias, K, mode = pack.pack_conv(kern, np.ones(Cout, np.float32), np.zeros(Cout, np.float32), 4)
x = torch.randn(B, Hh, Ww, 3, device="cuda")
OH = OW = 112
pt = pl = 3
kwp = (k + 1) // 2 * 2
wp = max(Ww + pl, (OW - 1) * stride + kwp); wp += wp & 1
hp = max(Hh + pt, (OH - 1) * stride + k)
xd = H.cast_input_pad(x, (pt, hp - Hh - pt, pl, wp - Ww - pl))
conv = dict(mode=1, B=B, H=hp, W=wp // 2, Cin=8, KH=k, KW=kwp // 2, stride=stride, stride_w=stride // 2,
            pad_t=0, pad_l=0, OH=OH, OW=OW)
wd, bd = H.dev_bits(wt), H.dev_f32(bias)
out = torch.empty(B * OH * OW, Cout, dtype=torch.bfloat16, device="cuda")
for hint in hints:
    for _ in range(3):
        H.gemm(xd, wd, Cout, K, bias=bd, act="relu", conv=conv, tile_hint=hint, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        H.gemm(xd, wd, Cout, K, bias=bd, act="relu", conv=conv, tile_hint=hint, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"hint {hint}: {e0.elapsed_time(e1) * 100:.1f} us  (K={K})")
    if stamps:
        b = buf.cpu().numpy().reshape(8, 64)
        for wv in (0, 5):
            t = b[wv][b[wv] > 0]
            print("  wave", wv, "stamps:", (t[1:] - t[:-1])[:40])

# fused stem + pooling vs the unfused pair
def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


conv_out = out.view(B, OH, OW, Cout)
print(f"maxpool alone: {timed(lambda: H.maxpool(conv_out, 3, 2, 1)):.1f} us")
print(f"fused stem_conv_pool: {timed(lambda: H.stem_conv_pool(xd, wd, bd, B, hp, wp // 2, OH, OW)):.1f} us")
