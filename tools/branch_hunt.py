"""Which launch gives another result when two plans run side by side?  Two half-batch plans (every tensor its own buffer),
reference = each plan launched eagerly by itself; then the two recorded as parallel branches of one HIP graph are replayed N
times and every op output of both plans is compared with its reference checksum.

    python tools/branch_hunt.py [model] [batch] [replays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd")):
    sys.path.insert(0, p)
import torch

import bench
import tfimm
from tfimm.engine.graph import CapturedBranches
from tfimm.utils.init import synthetic_weights


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cait_xxs24_224"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = bench.synthetic_batch(model.cfg, B, 11)
    prog = model.program()
    for t in prog.tensors:
        t.keep = True
    plans = prog.make_branches(B, 2)
    ops = [op for op in prog.ops if op.output is not None and prog.tensors[op.output].dtype != "raw"]

    def sums(plan):
        out = []
        for op in ops:
            v = plan.tensor_view(prog.tensors[op.output])
            bits = v.reshape(-1).view(torch.int16 if v.dtype == torch.bfloat16 else torch.int32)
            out.append(int(bits.sum(dtype=torch.int64).item()))
        return out

    lo = 0
    ref = []
    for p in plans:
        p.run(x[lo:lo + p.batch])
        torch.cuda.synchronize()
        ref.append(sums(p))
        lo += p.batch
    eager = os.environ.get("EAGER", "0") == "1"       # two streams, launches one by one (no graph)
    cap = None if eager else CapturedBranches(plans, x)
    streams = [torch.cuda.Stream() for _ in plans]
    first_bad = {}
    for r in range(n):
        if eager:
            lo = 0
            import threading
            def go(p, s, lo):
                with torch.cuda.stream(s):
                    p.run(x[lo:lo + p.batch])
            ths = []
            for p, s in zip(plans, streams):
                ths.append(threading.Thread(target=go, args=(p, s, lo)))
                lo += p.batch
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
        else:
            cap.replay()
        torch.cuda.synchronize()
        for pi, p in enumerate(plans):
            cur = sums(p)
            diff = [i for i, (a, b) in enumerate(zip(ref[pi], cur)) if a != b]
            if diff:
                op = ops[diff[0]]
                first_bad.setdefault((diff[0], op.kind), 0)
                first_bad[(diff[0], op.kind)] += 1
                print(f"replay {r} plan {pi}: {len(diff)} outputs differ, first = op {diff[0]} {op.kind} {op.cite} "
                      f"{ {k: v for k, v in op.attrs.items() if k in ('M', 'N', 'K', 'heads', 'hd', 'n_tokens', 'rows', 'd', 'act')} }", flush=True)
    print(f"{name} B={B}: first-differing ops over {n} replays: {first_bad or 'none'}")


if __name__ == "__main__":
    main()
