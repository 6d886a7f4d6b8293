"""Run one GEMM shape repeatedly (for rocprofv3 --pmc / kernel-trace):  gemm_probe.py M K N hint [iters] [residual] [act]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import hip_ops as H
M, K, N, hint = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
Kp = (K + 63) // 64 * 64     # weights are stored zero-padded to whole 64-wide k-tiles (engine/pack.py)
w = torch.zeros(N, Kp, device="cuda", dtype=torch.bfloat16)
w[:, :K] = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").to(torch.bfloat16) if len(sys.argv) > 6 and sys.argv[6] not in ("0", "") else None
act = sys.argv[7] if len(sys.argv) > 7 else ""
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(2):
    H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=("table" if hint == 0 else hint), residual=res, act=act, act_after_res=res is not None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=("table" if hint == 0 else hint), residual=res, act=act, act_after_res=res is not None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
gb = (M * K + M * N * (2 if res is not None else 1)) * 2 / 1e9
print(f"M={M} K={K} N={N} hint={hint} act={act or '-'}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF/s {gb/ms*1e3:.0f} GB/s")
