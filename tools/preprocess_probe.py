"""Time the input conversion of a B x 224 x 224 x 3 batch: float32 cast vs fused uint8 preprocessing.
   preprocess_probe.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import hip_ops as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
u = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, device="cuda")
f = u.float()
mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


npx = B * 224 * 224
for name, fn, bytes_in in [
        ("cast_input fp32 3->4", lambda: H.cast_input(f, 4), 12),
        ("preprocess u8 3->4", lambda: H.preprocess_input(u, 4, mean, std), 3),
        ("cast_input_pad fp32 (pad 3)", lambda: H.cast_input_pad(f, (3, 3, 3, 3)), 12),
        ("preprocess_pad u8 (pad 3)", lambda: H.preprocess_input_pad(u, (3, 3, 3, 3), mean, std), 3)]:
    us = timed(fn)
    print(f"{name:32s} {us:8.1f} us  {(npx * (bytes_in + 8)) / us / 1e3:7.0f} GB/s")
