"""The input-strip convolution (tile hint 31) at the scored shape, many launches, bit for bit against the implicit-GEMM tile."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, R + "/tensorflow-image-models_amd", R + "/tests"]
import numpy as np, torch
import hip_ops as H
from tfimm.engine import pack
r = np.random.default_rng(0)
C = 128
kern = (r.standard_normal((3, 3, C, C)) / 34).astype(np.float32)
wt, bias, K, mode = pack.pack_conv(kern, np.ones(C, np.float32), r.standard_normal(C).astype(np.float32), C)
wd, bd = H.dev_bits(wt), H.dev_f32(bias)
for B in (256, 128, 2):
    x = torch.randn(B * 784, C, device="cuda").to(torch.bfloat16)
    conv = dict(mode=mode, B=B, H=28, W=28, Cin=C, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, OH=28, OW=28)
    ref = H.gemm(x, wd, C, K, bias=bd, act="relu", conv=conv, tile_hint=24).view(torch.int16).clone()
    bad = 0
    for rep in range(40):
        out = H.gemm(x, wd, C, K, bias=bd, act="relu", conv=conv, tile_hint=31)
        torch.cuda.synchronize()
        d = out.view(torch.int16) != ref
        if bool(d.any().item()):
            bad += 1
            if bad <= 3:
                rows = torch.nonzero(d.any(dim=1)).flatten().cpu().numpy()
                cols = torch.nonzero(d.any(dim=0)).flatten().cpu().numpy()
                print(f"  B={B} rep {rep}: {int(d.sum())} elements; rows {rows[:6].tolist()} ({rows.size}) rows%128 {sorted(set((rows % 128).tolist()))[:8]}; cols {cols.min()}..{cols.max()} ({cols.size})", flush=True)
    print(f"B={B}: {bad} / 40 launches differ from the implicit-GEMM tile", flush=True)
