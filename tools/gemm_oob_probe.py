"""Does a GEMM launch write anything outside its output tensor?  All operands are carved out of ONE arena filled with a
sentinel, with gaps between them; after the launch every byte outside the output must still hold the sentinel (or the
operand's own data).     python tools/gemm_oob_probe.py [M] [K] [N]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import hip_ops as H
from tfimm.engine import pack

M = int(sys.argv[1]) if len(sys.argv) > 1 else 6272
K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
N = int(sys.argv[3]) if len(sys.argv) > 3 else 192
r = np.random.default_rng(0)
GAP = 1 << 20
sizes = dict(a=M * K * 2, res=M * N * 2, out=M * N * 2)
arena = torch.full((sum(sizes.values()) + 4 * GAP,), 0x5A, dtype=torch.uint8, device="cuda")
off, views = GAP, {}
for k_, n in sizes.items():
    views[k_] = arena[off:off + n]
    off += n + GAP
a = views["a"].view(torch.bfloat16).view(M, K)
a.copy_(torch.randn(M, K, device="cuda").to(torch.bfloat16))
res = views["res"].view(torch.bfloat16).view(M, N)
res.copy_(torch.randn(M, N, device="cuda").to(torch.bfloat16))
out = views["out"].view(torch.bfloat16).view(M, N)
wt, _ = pack.pack_dense((r.standard_normal((K, N)) / math.sqrt(K)).astype(np.float32), None)
wd, b = H.dev_bits(wt), H.dev_f32(r.standard_normal(N).astype(np.float32))
snap = arena.clone()
o0 = views["out"].data_ptr() - arena.data_ptr()
for hint in (21, 22, 23, 24, 25, 26, 27, 29, 30, 11, 13, 1):
    arena.copy_(snap)
    H.gemm(a, wd, N, K, bias=b, residual=res, act="", tile_hint=hint, out=out)
    H.sync()
    d = (arena != snap)
    d[o0:o0 + sizes["out"]] = False
    n = int(d.sum().item())
    where = torch.nonzero(d).flatten()
    msg = ""
    if n:
        lo, hi = int(where[0].item()), int(where[-1].item())
        msg = f": bytes {lo} .. {hi} of the arena (output occupies {o0} .. {o0 + sizes['out']})"
    print(f"hint {hint}: {n} bytes outside the output changed{msg}", flush=True)
