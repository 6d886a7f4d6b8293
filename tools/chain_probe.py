"""Fused bottleneck tail vs the two-launch path on the ResNet-50 shapes: bit-equality of the outputs and time.
Usage (GPU box): python tools/chain_probe.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
import math  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hip_ops as H  # noqa: E402
from tfimm.engine import pack  # noqa: E402


def run(B, HW, C1, N2, stride, iters=20):
    r = np.random.default_rng(1)
    x = H.dev_bf16(r.standard_normal((B, HW, HW, C1)).astype(np.float32))
    k1 = (r.standard_normal((3, 3, C1, C1)) / math.sqrt(9 * C1)).astype(np.float32)
    k2 = (r.standard_normal((1, 1, C1, N2)) / math.sqrt(C1)).astype(np.float32)
    t1, t2 = r.standard_normal(C1).astype(np.float32), r.standard_normal(N2).astype(np.float32)
    wt1, b1, K1, mode = pack.pack_conv(k1, None, t1, C1)
    wt2, b2 = pack.pack_dense(k2.reshape(C1, N2)[pack.chain_k_order(C1)], t2)
    wt2u, _ = pack.pack_dense(k2.reshape(C1, N2), t2)
    OH = (HW + 2 - 3) // stride + 1
    res = H.dev_bf16(r.standard_normal((B * OH * OH, N2)).astype(np.float32))
    d1, d2, db1, db2 = H.dev_bits(wt1), H.dev_bits(wt2), H.dev_f32(b1), H.dev_f32(b2)
    d2u = H.dev_bits(wt2u)
    conv = dict(mode=mode, B=B, H=HW, W=HW, Cin=C1, KH=3, KW=3, stride=stride, pad_t=1, pad_l=1, OH=OH, OW=OH)

    def two():
        mid = H.gemm(x, d1, C1, K1, bias=db1, act="relu", conv=conv)
        return H.gemm(mid, d2u, N2, C1, bias=db2, residual=res, act="relu", act_after_res=True)

    def one():
        return H.conv_chain(x, d1, db1, d2, db2, res, KH=3, KW=3, stride=stride, pad=1, OH=OH, OW=OH, C1=C1, N2=N2)

    a, b = two(), one()
    torch.cuda.synchronize()
    same = torch.equal(a.view(-1), b.view(-1))
    diff = float((a.float().view(-1) - b.float().view(-1)).abs().max())
    ts = []
    for fn in (two, one):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    M = B * OH * OH
    byt = (B * HW * HW * C1 + 2 * M * N2) * 2
    fl = 2.0 * M * (9 * C1 * C1 + C1 * N2)
    print(f"B={B} {HW}x{HW} C1={C1} N2={N2} s={stride}: two launches {ts[0]:7.1f} us, fused {ts[1]:7.1f} us "
          f"({byt / ts[1] / 1e6:5.2f} TB/s, {fl / ts[1] / 1e6:6.1f} TF/s)  bit-equal={same} max|diff|={diff:.3g}")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    run(B, 56, 64, 256, 1)
