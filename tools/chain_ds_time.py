"""The fused bottleneck tail with the shortcut convolution inside (first block of ResNet-50 stage 1) at the scored shape:
time per launch and a bit checksum of the output -- run once per library to compare two builds
(TFIMM_HIP_LIB=... python tools/chain_ds_time.py [batch] [plain]).  `plain`: the flavour with a residual tensor instead."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hip_ops as H  # noqa: E402
from tfimm.engine import pack  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    plain = len(sys.argv) > 2 and sys.argv[2] == "plain"
    r = np.random.default_rng(5)
    C1, N2, HW = 64, 256, 56
    y1 = H.dev_bf16(r.standard_normal((B, HW, HW, C1)).astype(np.float32))
    x0 = H.dev_bf16(r.standard_normal((B * HW * HW, 64)).astype(np.float32))
    k1 = (r.standard_normal((3, 3, C1, C1)) / math.sqrt(9 * C1)).astype(np.float32)
    k2 = (r.standard_normal((C1, N2)) / math.sqrt(C1)).astype(np.float32)
    kd = (r.standard_normal((64, N2)) / 8).astype(np.float32)
    t1, t2 = r.standard_normal(C1).astype(np.float32), r.standard_normal(N2).astype(np.float32)
    wt1, b1, _, _ = pack.pack_conv(k1, None, t1, C1)
    wt2, b2 = pack.pack_dense(k2[pack.chain_k_order(C1)], t2)
    d1, db1, d2, db2, dds = H.dev_bits(wt1), H.dev_f32(b1), H.dev_bits(wt2), H.dev_f32(b2), H.dev_bits(pack.pack_chain_ds(kd))
    res = H.dev_bf16(r.standard_normal((B * HW * HW, N2)).astype(np.float32)) if plain else None

    def one():
        if plain:
            return H.conv_chain(y1, d1, db1, d2, db2, res, KH=3, KW=3, stride=1, pad=1, OH=HW, OW=HW, C1=C1, N2=N2)
        return H.conv_chain(y1, d1, db1, d2, db2, None, KH=3, KW=3, stride=1, pad=1, OH=HW, OW=HW, C1=C1, N2=N2, ds_x=x0, ds_w=dds)

    outs = [one() for _ in range(3)]
    torch.cuda.synchronize()
    sums = [int(o.view(torch.int16).to(torch.int64).sum().item()) for o in outs]
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            one()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    print(f"{os.environ.get('TFIMM_HIP_LIB', 'product library'):60s} {'plain' if plain else 'shortcut conv inside'} B={B}: "
          f"{min(ts):7.1f} us (median {sorted(ts)[2]:.1f})  checksum {sums[0]}  launches equal {len(set(sums)) == 1}")


if __name__ == "__main__":
    main()
