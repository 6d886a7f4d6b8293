"""NEEDS A PROBE BUILD: the *_DBG switches / stamps exist only with -DTFIMM_PROBE_HOOKS (tools/probes/build_dbg_libs.sh all; TFIMM_HIP_LIB=tools/probes/bin/libtfimm_hip_probe.so).
Dump s_memtime stamps of workgroup 0 of the persistent GEMM: pp_stamps.py M K N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
os.environ["TFIMM_GEMM_DBG"] = str(64 + (int(sys.argv[4]) if len(sys.argv) > 4 else 0))
os.environ["TFIMM_GEMM_DBG_PTR"] = hex(buf.data_ptr())
import hip_ops as H
M, K, N = (int(v) for v in sys.argv[1:4])
HINT = int(sys.argv[5]) if len(sys.argv) > 5 else 21
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(20):
    H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=HINT)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=HINT); e1.record(); torch.cuda.synchronize()
print("kernel us", e0.elapsed_time(e1) * 1e3)
b = buf.cpu().numpy().reshape(8, 64)
for wv in (0, 5):
    t = b[wv][b[wv] > 0]
    d = (t[1:] - t[:-1])
    print("wave", wv, "n", len(t), "total ticks", t[-1] - t[0])
    print("  per tile [kloop, p0 writes, p0, p1, p2, p3, barrier+next start]:")
    for i in range(0, len(d) - 6, 7):
        print("   ", d[i:i + 7])
