"""Run the depthwise conv repeatedly:  dw_probe.py B H C k stride [sums] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import hip_ops as H
B, Hh, C, k, s = (int(v) for v in sys.argv[1:6])
sums = int(sys.argv[6]) if len(sys.argv) > 6 else 0
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
x = torch.randn(B, Hh, Hh, C, device="cuda").to(torch.bfloat16)
w = torch.randn(k * k, C, device="cuda")
b = torch.randn(C, device="cuda")
pad = k // 2
OH = (Hh + 2 * pad - k) // s + 1
for _ in range(2):
    H.dwconv(x, w, b, k, s, pad, pad, OH, OH, act="swish", want_sums=bool(sums))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    H.dwconv(x, w, b, k, s, pad, pad, OH, OH, act="swish", want_sums=bool(sums))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
byts = B * (Hh * Hh + OH * OH) * C * 2
print(f"dwconv B={B} H={Hh} C={C} k={k} s={s} sums={sums}: {ms*1e3:.1f} us {byts/ms/1e6:.0f} GB/s")
