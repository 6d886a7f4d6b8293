"""Do two half-batch replays on two HIP streams fill each other's idle CUs (under-filled launches, launch tails)?
Times, for one workload: the full-batch graph; the two half-batch graphs one after the other on one stream; the two half-batch
graphs side by side on two streams.      python tools/two_stream_probe.py [workload] [batch] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd")):
    sys.path.insert(0, p)
import torch

import bench
import tfimm
from tfimm.utils.init import synthetic_weights


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS[name]["batch"]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    parts = int(os.environ.get("PARTS", "2"))
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = bench.synthetic_batch(model.cfg, B, 2021)
    prog = model.program()
    full = prog.make_plan(B).capture(x)
    hb = B // parts
    halves = [prog.make_plan(hb).capture(x[i * hb:(i + 1) * hb]) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    def seq():
        for h in halves:
            h.replay()

    def par():
        cur = torch.cuda.current_stream()
        for s, h in zip(streams, halves):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                h.replay()
        for s in streams:
            cur.wait_stream(s)

    def free_running(skew_ms):
        """no join between steps: every stream replays its slice back to back, the second one started ``skew_ms`` late"""
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, (s, h) in enumerate(zip(streams, halves)):
            with torch.cuda.stream(s):
                if k and skew_ms > 0:
                    torch.cuda._sleep(int(skew_ms * 1e-3 * 2.0e9 * k))
        for _ in range(iters):
            for s, h in zip(streams, halves):
                with torch.cuda.stream(s):
                    h.replay()
        torch.cuda.synchronize()
        return ((time.perf_counter() - t0) * 1e3 - skew_ms * (parts - 1)) / iters

    t_full, t_seq, t_par = timed(full.replay), timed(seq), timed(par)
    for skew in (0.0, 0.25, 0.5):
        free_running(skew * t_full)
        print(f"   free-running streams, second one {skew:.2f} of a forward late: {free_running(skew * t_full):.3f} ms per step")
    print(f"{name} B={B}: full-batch graph {t_full:.3f} ms; {parts} x B={hb} one stream {t_seq:.3f} ms; {parts} streams {t_par:.3f} ms "
          f"({B / t_par:.1f} k img/s vs {B / t_full:.1f})")


if __name__ == "__main__":
    main()
