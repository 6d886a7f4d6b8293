"""Run the talking-heads attention kernel repeatedly (timing / rocprofv3 --pmc):  tha_probe.py B N heads hd [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import hip_ops as H
B, N, heads, hd = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
r = np.random.default_rng(0)
qkv = torch.randn(B * N, 3 * heads * hd, device="cuda").to(torch.bfloat16)
wl = (np.eye(heads) + 0.1 * r.standard_normal((heads, heads))).astype(np.float32)
ww = (np.eye(heads) + 0.1 * r.standard_normal((heads, heads))).astype(np.float32)
bl = np.zeros(heads, np.float32); bw = np.zeros(heads, np.float32)
for _ in range(2):
    H.talking_heads_attention(qkv, B, N, heads, hd, hd ** -0.5, wl, bl, ww, bw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    H.talking_heads_attention(qkv, B, N, heads, hd, hd ** -0.5, wl, bl, ww, bw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 4.0 * B * heads * N * N * hd
print(f"B={B} N={N} heads={heads} hd={hd}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s (QK^T + PV flops only)")
