"""Find the launch whose output is not reproducible: every tensor of the layer program keeps its own buffer (no slab reuse),
the recorded graph is replayed N times (and the plan launched eagerly a few times), and after every run the bit checksum of
EVERY op output is compared with the first run's.  Prints the first op (in program order) that ever differed.

    python tools/flaky_hunt.py [model] [batch] [replays] [eager runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorflow-image-models_amd")):
    sys.path.insert(0, p)
import torch

import bench
import tfimm
from tfimm.utils.init import synthetic_weights


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet_b4"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    n_replay = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    n_eager = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    model = tfimm.create_model(name)
    model.set_weights(synthetic_weights(model, 2021))
    x = bench.synthetic_batch(model.cfg, B, 7)
    prog = model.program()
    for t in prog.tensors:
        t.keep = True
    plan = prog.make_plan(B)
    ops = [op for op in prog.ops if op.output is not None]

    def sums():
        out = []
        for op in ops:
            t = prog.tensors[op.output]
            if t.dtype == "raw":
                out.append(0)
                continue
            v = plan.tensor_view(t)
            bits = v.reshape(-1).view(torch.int16 if v.dtype == torch.bfloat16 else torch.int32)
            out.append(int(bits.sum(dtype=torch.int64).item()))
        return out

    plan.run(x)
    torch.cuda.synchronize()
    ref = sums()
    bad = {}
    small = {}       # copies of the small tensors of the first run (inputs / extra outputs of a differing op are compared by value)
    for t in prog.tensors:
        if t.dtype != "raw" and t.rows * t.C * B <= (1 << 24):
            small[t.id] = plan.tensor_view(t).clone()

    def explain(op):
        for tid in list(op.inputs) + [op.output] + list(op.extra_outputs):
            t = prog.tensors[tid]
            if tid in small:
                cur, old = plan.tensor_view(t), small[tid]
                n = int((cur.view(torch.int32) != old.view(torch.int32)).sum().item()) if cur.dtype == torch.float32 else \
                    int((cur.view(torch.int16) != old.view(torch.int16)).sum().item())
                extra = ""
                if n and cur.dtype == torch.float32 and "sums" in (t.name or ""):
                    d = (cur.reshape(B, -1).view(torch.int64) - old.reshape(B, -1).view(torch.int64))
                    nz = torch.nonzero(d)
                    extra = f" int64 diffs: {d[d != 0][:6].tolist()} at (image, channel) {nz[:6].tolist()}"
                print(f"      tensor {tid} '{t.name}' rows={t.rows} C={t.C} {t.dtype}: {n} words differ from run 0{extra}")
    def check(tag):
        torch.cuda.synchronize()
        cur = sums()
        diff = [i for i, (a, b) in enumerate(zip(ref, cur)) if a != b]
        if diff:
            i = diff[0]
            bad.setdefault(i, []).append(tag)
            op = ops[i]
            print(f"{tag}: {len(diff)} outputs differ, first = op {i} {op.kind} {op.cite} "
                  f"{ {k: v for k, v in op.attrs.items() if k in ('M', 'N', 'K', 'C', 'k', 'stride', 'H', 'W', 'OH', 'Cin', 'act', 'mode')} }", flush=True)
            if len(bad[i]) <= 2:
                explain(op)
                if i > 0:
                    explain(ops[i - 1])
    for e in range(n_eager):
        plan.run(x)
        check(f"eager {e}")
    cap = plan.capture(x)
    for r in range(n_replay):
        cap.replay()
        check(f"replay {r}")
    print(f"{name} B={B}: {len(ops)} op outputs, {n_eager} eager + {n_replay} replays; first-differing ops: "
          f"{ {i: (ops[i].kind, len(v)) for i, v in bad.items()} or 'none'}")


if __name__ == "__main__":
    main()
