"""Run the attention op repeatedly:  attn_probe.py B N heads hd [window res shift] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import hip_ops as H
B, N, heads, hd = (int(v) for v in sys.argv[1:5])
window = int(sys.argv[5]) if len(sys.argv) > 5 else 0
res = int(sys.argv[6]) if len(sys.argv) > 6 else 0
shift = int(sys.argv[7]) if len(sys.argv) > 7 else 0
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
qkv = torch.randn(B * N, 3 * heads * hd, device="cuda").to(torch.bfloat16)
kw = dict(window=window, shift=shift, res=(res, res)) if window else {}
if window:
    import numpy as np
    from tfimm.engine import pack
    rb = np.random.default_rng(0).standard_normal((heads, window * window, window * window)).astype(np.float32)
    kw["rel_bias"] = torch.from_numpy(rb).cuda()
    if os.environ.get("ATTN_NO_TILES", "0") != "1":
        kw["bias_log2"] = torch.from_numpy(pack.swin_bias_tiles(rb, window, shift)).cuda()
for _ in range(2):
    H.attention(qkv, B, N, heads, hd, hd ** -0.5, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    H.attention(qkv, B, N, heads, hd, hd ** -0.5, **kw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
n = window * window if window else N
flops = 4.0 * B * (N // n) * heads * n * n * hd
print(f"attn B={B} N={N} h={heads} hd={hd} win={window}: {ms*1e3:.1f} us {flops/ms/1e9:.1f} TF/s {B*N*heads*hd*2*4/ms/1e6:.0f} GB/s")
