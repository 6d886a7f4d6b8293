"""tfimm_hip_talking_heads_attention on many images (several workgroups per CU): reproducible from launch to launch, and equal
to the same images run one at a time?     python tools/tha_repro_probe.py [batch] [heads] [hd] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import hip_ops as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hd = int(sys.argv[3]) if len(sys.argv) > 3 else 48
N = int(sys.argv[4]) if len(sys.argv) > 4 else 196
r = np.random.default_rng(1)
g = torch.Generator(device="cuda").manual_seed(2)
qkv = torch.randn(B * N, 3 * heads * hd, device="cuda", generator=g).to(torch.bfloat16)
wl = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
ww = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
bl = (0.3 * r.standard_normal(heads)).astype(np.float32)
bw = (0.02 * r.standard_normal(heads)).astype(np.float32)
first = None
for run in range(6):
    out = H.talking_heads_attention(qkv, B, N, heads, hd, hd ** -0.5, wl, bl, ww, bw)
    H.sync()
    o = out.view(torch.int16)
    if first is None:
        first = o.clone()
    else:
        d = (o != first)
        n = int(d.sum().item())
        if n:
            rows = torch.nonzero(d.any(dim=1)).flatten()
            print(f"run {run}: {n} elements differ from run 0; images {sorted(set((rows // N).tolist()))[:12]} rows-in-image {sorted(set((rows % N).tolist()))[:16]}")
one = torch.cat([H.talking_heads_attention(qkv[i * N:(i + 1) * N].contiguous(), 1, N, heads, hd, hd ** -0.5, wl, bl, ww, bw) for i in range(min(B, 8))])
H.sync()
d = (one.view(torch.int16) != first[: one.shape[0]])
print(f"B={B} heads={heads} hd={hd} n={N}: first 8 images one at a time vs in the batch: {int(d.sum().item())} elements differ")
