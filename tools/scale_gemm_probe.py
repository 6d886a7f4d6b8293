"""SE-gated projection GEMMs (a_scale) of EfficientNet-B4 at batch 256, one shape per row, every tile hint a column: us per
launch (best of `rounds` interleaved passes of `iters` launches), so that the table entries of those shapes can be chosen from
numbers instead of from one noisy tuner pass.

    python tools/scale_gemm_probe.py [iters] [rounds] [batch]      -> gpurun_out/scale_gemm_probe[_b<batch>].txt
(batch 128 = the shapes of the two half-batch branches bench.py replays)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hip_ops as H  # noqa: E402
from tfimm.engine import tune  # noqa: E402

# (K, N, rows per image, residual); M = batch x rows per image
LAYERS = [(1632, 272, 144, True), (1632, 448, 144, False), (2688, 448, 144, True), (960, 272, 144, False),
          (960, 160, 576, True), (672, 160, 576, False), (672, 112, 576, True), (336, 112, 576, False),
          (336, 56, 2304, True), (192, 56, 2304, False), (192, 32, 9025, True), (144, 32, 9025, False)]
HINTS = [0, 21, 22, 23, 24, 25, 26, 27, 29, 30, 1, 2, 3, 4, 5, 6]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    SHAPES = [(batch * R, K, N, R, res) for K, N, R, res in LAYERS]
    lines = ["shape (M K N rows/img res)".ljust(34) + "table " + " ".join(f"{h:>6d}" for h in HINTS)]
    for M, K, N, R, has_res in SHAPES:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        Kp = (K + 63) // 64 * 64
        w = torch.zeros(N, Kp, device="cuda", dtype=torch.bfloat16)
        w[:, :K] = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        g = torch.rand(M // R, K, device="cuda")
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda").to(torch.bfloat16) if has_res else None
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        best = {h: float("inf") for h in HINTS}
        ref = None
        table_hint = None
        for _ in range(rounds):
            for h in HINTS:
                def run():
                    H.gemm(a, w, N, K, bias=bias, out=out, tile_hint=h, residual=res, a_scale=g, rows_per_image=R)
                try:
                    run()
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = out.clone()
                    elif not torch.equal(ref, out):            # every tile must give the same bits up to accumulation order
                        err = float((ref.float() - out.float()).abs().max() / (ref.float().abs().max() + 1e-6))
                        assert err < 2e-2, (M, K, N, h, err)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _i in range(iters):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    best[h] = min(best[h], e0.elapsed_time(e1) / iters * 1e3)
                except RuntimeError:
                    best[h] = float("nan")
        d = H.ffi.GemmDesc()
        d.mode, d.M, d.N, d.K, d.lda, d.ldc = 0, M, N, K, K, N
        d.residual = H.ptr(res)
        d.a_scale = H.ptr(g)
        table_hint = tune.lookup(d)
        lines.append(f"{M:8d} {K:5d} {N:4d} {R:5d} {int(has_res)}".ljust(34) + f"{table_hint:5d} " +
                     " ".join(f"{best[h]:6.1f}" for h in HINTS))
        print(lines[-1], flush=True)
        del a, w, g, out, res
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "scale_gemm_probe.txt" if batch == 256 else f"scale_gemm_probe_b{batch}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    print("shape (M K N rows/img res)".ljust(34) + "table " + " ".join(f"{h:>6d}" for h in HINTS))
    main()
