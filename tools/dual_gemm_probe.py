"""The three GEMMs of ResNet-50 that carry their block's shortcut convolution as a second A operand (tfimm_gemm_desc::a2), every
persistent tile hint a column: us per launch (best of `rounds` interleaved passes) and the table key of each shape.

    python tools/dual_gemm_probe.py [iters] [rounds] [batch ...]      -> gpurun_out/dual_gemm_probe.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ops as H  # noqa: E402
from tfimm.engine import tune  # noqa: E402

# (output size, K1, K2, N, stride)
LAYERS = [(28, 128, 256, 512, 2), (14, 256, 512, 1024, 2), (7, 512, 1024, 2048, 2)]
HINTS = list(tune.DUAL_CANDIDATES)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    batches = [int(v) for v in sys.argv[3:]] or [256, 128]
    lines = ["batch out K1 K2 N".ljust(28) + "table " + " ".join(f"{h:>6d}" for h in HINTS) + "   key"]
    for batch in batches:
        for O, K1, K2, N, s in LAYERS:
            M, Hin = batch * O * O, O * s
            h = torch.randn(M, K1, device="cuda").to(torch.bfloat16)
            x = torch.randn(batch * Hin * Hin, K2, device="cuda").to(torch.bfloat16)
            k1p, k2p = -(-K1 // 64) * 64, -(-K2 // 64) * 64
            w = torch.zeros(N, k1p + k2p, device="cuda", dtype=torch.bfloat16)
            w[:, :K1] = (torch.randn(N, K1, device="cuda") / K1 ** 0.5).to(torch.bfloat16)
            w[:, k1p:k1p + K2] = (torch.randn(N, K2, device="cuda") / K2 ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda")
            out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            geom = (s, Hin, Hin, O, O)
            best = {hh: float("inf") for hh in HINTS}
            for _ in range(rounds):
                for hh in HINTS:
                    def run():
                        H.gemm(h, w, N, K1, bias=bias, out=out, act="relu", tile_hint=hh, a2=x, a2_geom=geom)
                    run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _i in range(iters):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    best[hh] = min(best[hh], e0.elapsed_time(e1) / iters * 1e3)
            d = H.ffi.GemmDesc()
            d.mode, d.M, d.N, d.K, d.lda, d.ldc, d.act = 0, M, N, K1, K1, N, H.ffi.ACT["relu"]
            d.a2, d.K2, d.a2_stride = H.ptr(x), K2, s
            lines.append(f"{batch:4d} {O:3d} {K1:5d} {K2:5d} {N:5d}".ljust(28) + f"{tune.lookup(d):5d} " +
                         " ".join(f"{best[hh]:6.1f}" for hh in HINTS) + "   " + tune.key_of(d))
            print(lines[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dual_gemm_probe.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
