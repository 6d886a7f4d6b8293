"""Which launch disturbs tfimm_hip_talking_heads_attention (heads 4, hd 48, 196 tokens, 32 images) when it runs next to it on
another stream?  For every candidate neighbour: the talking-heads launch runs N times on stream A while the neighbour is
launched back to back on stream B; every talking-heads result is compared bit for bit with a solo run.
    python tools/tha_neighbour_probe.py [repeats]"""
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

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, N, heads, hd = 32, 196, 4, 48
r = np.random.default_rng(1)
g = torch.Generator(device="cuda").manual_seed(2)
qkv = torch.randn(B * N, 3 * heads * hd, device="cuda", generator=g).to(torch.bfloat16)
wl = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
ww = (r.standard_normal((heads, heads)) / heads ** 0.5 + np.eye(heads)).astype(np.float32)
bl = (0.3 * r.standard_normal(heads)).astype(np.float32)
bw = (0.02 * r.standard_normal(heads)).astype(np.float32)


def tha():
    return H.talking_heads_attention(qkv, B, N, heads, hd, hd ** -0.5, wl, bl, ww, bw)


ref = tha().view(torch.int16).clone()
H.sync()
M = B * N


def dense(K, Nn, hint, act="", residual=False, ln=False, rows=None):
    M = rows or B * N
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (r.standard_normal((K, Nn)) / math.sqrt(K)).astype(np.float32)
    wt, _ = pack.pack_dense(w, None)
    wd, b = H.dev_bits(wt), H.dev_f32(r.standard_normal(Nn).astype(np.float32))
    res = torch.randn(M, Nn, device="cuda", generator=g).to(torch.bfloat16) if residual else None
    st = H.row_stats(a, 1e-6) if ln else None
    c1 = H.dev_bits(pack.pack_ln_c1(wt, Nn, K)) if ln else None
    out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
    return lambda: H.gemm(a, wd, Nn, K, bias=b, residual=res, act=act, tile_hint=hint, ln_stats=st, ln_c1=c1, out=out)


x_rs = torch.randn(M, 192, device="cuda", generator=g).to(torch.bfloat16)
CASES = [("nothing", None)]
for hint, nm in ((23, "128x128"), (24, "256x64"), (25, "128x64"), (27, "256x64 4x1"), (21, "256x256"), (30, "two-workgroup")):
    CASES += [(f"dense 192->576 LN-folded, tile {nm}", dense(192, 576, hint, ln=True)),
              (f"dense 192->192 +residual, tile {nm}", dense(192, 192, hint, residual=True)),
              (f"dense 192->768 gelu (no LN), tile {nm}", dense(192, 768, hint, act="gelu")),
              (f"dense 768->192 +residual, tile {nm}", dense(768, 192, hint, residual=True))]
CASES += [("row_stats", lambda: H.row_stats(x_rs, 1e-6)), ("talking heads itself", tha)]
if os.environ.get("BISECT", "0") == "1":          # variations of the one neighbour that disturbs: 768 -> 192 + residual on the 256x64 (4x1) tile
    CASES = [("nothing", None)]
    for K_, N_, res_, act_, rows_ in ((768, 192, True, "", 65536), (1536, 192, True, "", None), (768, 64, True, "", None)):
        CASES.append((f"hint 27: K={K_} N={N_} residual={res_} act={act_ or '-'} rows={rows_ or B * N}",
                      dense(K_, N_, 27, act=act_, residual=res_, rows=rows_)))

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name, nb in CASES:
    if nb is not None:
        nb()                      # lazy initialisation outside the measurement
    H.sync()
    bad = 0
    for rep in range(REP):
        with torch.cuda.stream(sb):
            if nb is not None:
                for _ in range(6):
                    nb()
        with torch.cuda.stream(sa):
            o = tha()
        torch.cuda.synchronize()
        dmask = (o.view(torch.int16) != ref)
        if bool(dmask.any().item()):
            bad += 1
            if bad == 1 and os.environ.get("BISECT", "0") == "1":
                rows = torch.nonzero(dmask.any(dim=1)).flatten()
                cols = torch.nonzero(dmask.any(dim=0)).flatten()
                mag = (o.float() - ref.view(torch.bfloat16).float()).abs().max().item()
                print(f"      {int(dmask.sum().item())} elements, max |diff| {mag:.3g}; {rows.numel()} rows: images {sorted(set((rows // N).tolist()))[:10]} "
                      f"queries {sorted(set((rows % N).tolist()))[:24]}; columns {cols[:8].tolist()} .. {cols[-4:].tolist()} ({cols.numel()})", flush=True)
    print(f"{name:55s} talking-heads results differing from the solo run: {bad} / {REP}", flush=True)
