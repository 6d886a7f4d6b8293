"""Per segment of the layer program: full-batch launches on one stream vs the two half-batch plans side by side (each segment
recorded into its own HIP graph; inputs are whatever the buffers hold -- timing only).  Shows where parallel branches pay.
    python tools/stage_fork_probe.py [model] [batch] [segments]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd")):
    sys.path.insert(0, p)
import torch

import bench
import tfimm
from tfimm.engine.graph import _hip_memset_async
from tfimm.utils.init import synthetic_weights


def run_range(plan, x, lo, hi):
    st_ptr = torch.cuda.current_stream().cuda_stream
    st = C.c_void_p(st_ptr)
    idx = plan._input_patch[0]
    for i in range(lo, hi):
        fn, args = plan.calls[i]
        if i == idx:
            plan.launch_input(x, st)
        elif fn == "memset":
            _hip_memset_async(args[0], args[1], st_ptr)
        else:
            rc = fn(*args, st)
            assert rc == 0, (i, rc)


def record(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    return g


def timed(g, n=10):
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS[name]["batch"]
    nseg = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = bench.synthetic_batch(model.cfg, B, 1)
    prog = model.program()
    full = prog.make_plan(B)
    halves = prog.make_branches(B, 2)
    for p, xs in ((full, x), (halves[0], x[:B // 2]), (halves[1], x[B // 2:])):
        p.run(xs)
    torch.cuda.synchronize()
    n = len(full.calls)
    assert all(len(h.calls) == n for h in halves)
    bounds = [round(i * n / nseg) for i in range(nseg + 1)]
    side = torch.cuda.Stream()
    tot_full = tot_fork = tot_best = 0.0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        gf = record(lambda: run_range(full, x, lo, hi))

        def fork():
            main_s = torch.cuda.current_stream()
            side.wait_stream(main_s)
            run_range(halves[0], x[:B // 2], lo, hi)
            with torch.cuda.stream(side):
                run_range(halves[1], x[B // 2:], lo, hi)
            main_s.wait_stream(side)
        gk = record(fork)
        tf, tk = timed(gf), timed(gk)
        tot_full += tf; tot_fork += tk; tot_best += min(tf, tk)
        kinds = [fn if isinstance(fn, str) else fn.__name__.replace("tfimm_hip_", "") for fn, _ in full.calls[lo:hi]]
        print(f"calls {lo:3d}..{hi:3d}: full {tf:7.3f} ms  forked {tk:7.3f} ms  ({tk / tf:5.2f})  {kinds[0]} .. {kinds[-1]}", flush=True)
    print(f"{name} B={B}: sum full {tot_full:.3f} ms, sum forked {tot_fork:.3f} ms, sum of per-segment best {tot_best:.3f} ms")


if __name__ == "__main__":
    main()
