"""A/B timing of GEMM tile hints on the layer shapes of the scored workloads (one GPU call, many shapes).

  python tools/duo_probe.py [vit|swin|resnet|all] [iters]

Every (shape, hint) is timed with HIP events over `iters` launches that cycle through NBUF distinct operand / output
buffers (so a launch does not find its own operands in L2 / Infinity Cache from the launch before).  Hint "table" = what
gemm_tune.json selects for the shape today.  Prints us, TF/s and algorithmic GB/s per variant.
"""
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

NBUF = 3


def dense(M, K, N, act="", res=False, ln=False, act_after_res=False):
    return dict(kind="dense", M=M, K=K, N=N, act=act, res=res, ln=ln, act_after_res=act_after_res)


def conv(B, HW, Cin, N, k, stride, act="relu", res=False):
    return dict(kind="conv", B=B, H=HW, W=HW, Cin=Cin, N=N, k=k, stride=stride, act=act, res=res)


SETS = {
    "vit": [dense(100864, 768, 2304, ln=True), dense(100864, 768, 768, res=True), dense(100864, 768, 3072, act="gelu", ln=True),
            dense(100864, 3072, 768, res=True)],
    "swin": [dense(802816, 128, 384, ln=True), dense(802816, 128, 128, res=True), dense(802816, 128, 512, act="gelu", ln=True),
             dense(802816, 512, 128, res=True),
             dense(200704, 256, 768, ln=True), dense(200704, 256, 1024, act="gelu", ln=True), dense(200704, 1024, 256, res=True),
             dense(50176, 512, 1536, ln=True), dense(50176, 512, 512, res=True), dense(50176, 512, 2048, act="gelu", ln=True),
             dense(50176, 2048, 512, res=True),
             dense(12544, 1024, 3072, ln=True), dense(12544, 1024, 4096, act="gelu", ln=True), dense(12544, 4096, 1024, res=True)],
    "resnet": [conv(256, 28, 128, 128, 3, 1), conv(256, 14, 256, 256, 3, 1), conv(256, 7, 512, 512, 3, 1),
               dense(802816, 256, 64, act="relu"), dense(802816, 256, 128, act="relu"),
               dense(200704, 512, 128, act="relu"), dense(200704, 128, 512, act="relu", res=True, act_after_res=True),
               dense(200704, 512, 256, act="relu"),
               dense(50176, 1024, 256, act="relu"), dense(50176, 256, 1024, act="relu", res=True, act_after_res=True),
               dense(50176, 1024, 512, act="relu"),
               dense(12544, 2048, 512, act="relu"), dense(12544, 512, 2048, act="relu", res=True, act_after_res=True)],
}
HINTS = ["table", 21, 22, 23, 28, 30]


def build(s):
    g = torch.Generator(device="cuda").manual_seed(1)
    if s["kind"] == "dense":
        M, K, N = s["M"], s["K"], s["N"]
        a = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(NBUF)]
        Kp = (K + 63) // 64 * 64
        w = torch.zeros(N, Kp, device="cuda", dtype=torch.bfloat16)
        w[:, :K] = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
        kw = dict(N=N, K=K, bias=torch.randn(N, device="cuda", generator=g), act=s["act"], act_after_res=s["act_after_res"])
        if s["ln"]:
            kw["ln_stats"] = H.row_stats(a[0], 1e-6)
            wt_host = w.cpu().view(torch.int16).numpy().view(np.uint16)
            kw["ln_c1"] = H.dev_bits(pack.pack_ln_c1(wt_host, N, K))
        res = [torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16) for _ in range(NBUF)] if s["res"] else None
        out = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(NBUF)]
        flops = 2.0 * M * N * K
        nbytes = (M * K + M * N * (2 if s["res"] else 1) + N * K) * 2
        def call(i, hint):
            H.gemm(a[i % NBUF], w, out=out[i % NBUF], residual=None if res is None else res[i % NBUF], tile_hint=hint, **kw)
        name = f"dense M={M} K={K} N={N}{' ln' if s['ln'] else ''}{' +res' if s['res'] else ''} {s['act'] or '-'}"
        return call, flops, nbytes, name, s["ln"]
    B, Hh, W, Cin, N, k, st = s["B"], s["H"], s["W"], s["Cin"], s["N"], s["k"], s["stride"]
    pad = k // 2
    OH = (Hh + 2 * pad - k) // st + 1
    M, K = B * OH * OH, k * k * Cin
    x = [torch.randn(B * Hh * W, Cin, device="cuda", generator=g).to(torch.bfloat16) for _ in range(NBUF)]
    Kp = (K + 63) // 64 * 64
    w = torch.zeros(N, Kp, device="cuda", dtype=torch.bfloat16)
    w[:, :K] = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    out = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(NBUF)]
    cv = dict(mode=1, B=B, H=Hh, W=W, Cin=Cin, KH=k, KW=k, stride=st, pad_t=pad, pad_l=pad, OH=OH, OW=OH)
    def call(i, hint):
        H.gemm(x[i % NBUF], w, N, K, bias=bias, act=s["act"], conv=cv, out=out[i % NBUF], tile_hint=hint)
    return call, 2.0 * M * N * K, (B * Hh * W * Cin + M * N + N * K) * 2, f"conv{k}x{k}s{st} M={M} K={K} N={N} {s['act']}", False


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    sets = list(SETS) if which == "all" else which.split(",")
    for key in sets:
        print(f"## {key}", flush=True)
        for s in SETS[key]:
            call, flops, nbytes, name, ln = build(s)
            row = []
            for hint in HINTS:
                if ln and hint == 28:
                    continue
                try:
                    for i in range(3):
                        call(i, hint)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(iters):
                        call(i, hint)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) / iters * 1e3
                    row.append(f"{hint}:{us:7.1f}us {flops / us / 1e6:6.0f}TF {nbytes / us / 1e3:5.0f}GB/s")
                except Exception as e:  # noqa: BLE001
                    row.append(f"{hint}: ERR {str(e)[:40]}")
            print(f"{name:58s} | " + " | ".join(row), flush=True)
            del call
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
