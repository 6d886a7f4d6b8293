import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
os.environ["TFIMM_GEMM_DBG"] = "64"
os.environ["TFIMM_GEMM_DBG_PTR"] = hex(buf.data_ptr())
import hip_ops as H
M, K, N = (int(v) for v in sys.argv[1:4])
act = sys.argv[4] if len(sys.argv) > 4 else ""
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(10):
    H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=21, act=act)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=21, act=act); e1.record(); torch.cuda.synchronize()
print(f"M={M} K={K} N={N} act={act}: kernel us", e0.elapsed_time(e1) * 1e3)
b = buf.cpu().numpy().reshape(8, 64)
for wv in (0, 5):
    t = b[wv][b[wv] > 0]
    d = (t[1:] - t[:-1])
    print(" wave", wv, "n", len(t), "total ticks", t[-1] - t[0], "diffs [kloop, epi+boundary]*:", list(d[:28]))
