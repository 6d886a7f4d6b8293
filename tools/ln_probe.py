"""Time LayerNorm over [rows, d] bf16:  ln_probe.py rows d [rows d ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import hip_ops as H
args = [int(v) for v in sys.argv[1:]] or [100864, 768]
for rows, d in zip(args[0::2], args[1::2]):
    x = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
    g, b = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
    for _ in range(3):
        H.layernorm(x, g, b, 1e-6)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        H.layernorm(x, g, b, 1e-6)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"layernorm rows={rows} d={d}: {us:.1f} us  {rows * d * 4 / us / 1e3:.0f} GB/s")
