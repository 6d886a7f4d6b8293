"""Debug helper: per-32x32-block error map of one GEMM for given tile hints."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import hip_ops as H
from tfimm.engine import pack

def run(M, K, N, hint, out_f32=False):
    r = np.random.default_rng(0)
    a = pack.bf16_bits_to_f32(pack.to_bf16_bits(r.standard_normal((M, K)).astype(np.float32))).reshape(M, K)
    w = pack.bf16_bits_to_f32(pack.to_bf16_bits((r.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32))).reshape(K, N)
    wt, _ = pack.pack_dense(w, None)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    got = H.gemm(H.dev_bf16(a), H.dev_bits(wt), N, K, tile_hint=hint, out_f32=out_f32)
    H.sync()
    g = got.float().cpu().numpy().astype(np.float64)
    err = np.abs(g - ref) / (np.abs(ref).max() + 1e-6)
    print(f"M={M} K={K} N={N} hint={hint} f32={out_f32} max err {err.max():.3e}")
    bm = (M + 31) // 32; bn = (N + 31) // 32
    for i in range(bm):
        row = ""
        for j in range(bn):
            e = err[i*32:(i+1)*32, j*32:(j+1)*32].max()
            row += "." if e < 1e-2 else "X"
        print("  ", row)

for hint in (22, 24, 26, 21):
    run(256, 64, 256, hint)
    run(256, 192, 320, hint)
run(256, 192, 320, 22, out_f32=True)
