"""Time tfimm_hip_attention:  attn_probe.py [batch n_tokens heads hd [window res]] ...  (default: ViT-B/16 at batch 512)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import hip_ops as H
cfgs = [(512, 197, 12, 64, 0, 0), (256, 196, 16, 32, 7, 14), (256, 3136, 4, 32, 7, 56), (512, 197, 6, 64, 0, 0), (256, 577, 12, 64, 0, 0)]
a = [int(v) for v in sys.argv[1:]]
if a:
    cfgs = [tuple(a[i:i + 6]) for i in range(0, len(a), 6)]
for B, n, heads, hd, win, res in cfgs:
    qkv = torch.randn(B * n, 3 * heads * hd, device="cuda").to(torch.bfloat16)
    kw = dict(window=win, shift=0, res=(res, res)) if win else {}
    for _ in range(3):
        H.attention(qkv, B, n, heads, hd, hd ** -0.5, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        H.attention(qkv, B, n, heads, hd, hd ** -0.5, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    nk = win * win if win else n
    fl = 4.0 * B * n * nk * heads * hd
    print(f"attention B={B} n={n} heads={heads} hd={hd} win={win}: {us:.1f} us  {fl / us / 1e6:.0f} TF/s  {B * n * heads * hd * 8 / us / 1e3:.0f} GB/s")
